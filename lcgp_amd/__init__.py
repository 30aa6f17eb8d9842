"""lcgp_amd -- MI355X-native hot path of the Latent Component Gaussian Process (LCGP) emulator.

Same public surface as the reference package (`src/lcgp/__init__.py:13`): LCGP, Matern32, test.
"""
from .lcgp import LCGP
from .covmat import Matern32, SquaredExponential
from . import evaluation

__version__ = "0.1.0"
__all__ = ['LCGP', 'Matern32', 'test']


def test(*args):
    """Runs the CPU-side test-suite of this repository (the reference's `lcgp.test()` hook, test.py:1-25)."""
    import os
    import pytest
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return pytest.main([os.path.join(here, 'tests'), '-q', '-m', 'not gpu', *args])
