"""Drop-in import shim: `from lcgp import LCGP, Matern32` resolves to the MI355X build (lcgp_amd).

Mirrors the export list of the reference package (`src/lcgp/__init__.py:13`)."""
from lcgp_amd import LCGP, Matern32, test, __version__  # noqa: F401
from lcgp_amd import evaluation  # noqa: F401

__all__ = ['LCGP', 'Matern32', 'test']
