"""NLL and gradient of three configurations through a given build of the library, saved as one .npy vector: two builds
that are meant to compute the same numbers are compared bit for bit with numpy.array_equal on the two files.

    python tools/dump_eval.py <path to liblcgp_hip.so variant> <out.npy>
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcgp_amd import _hip  # noqa: E402

_hip.LIB_PATH = sys.argv[1]
from lcgp_amd import LCGP, synth  # noqa: E402

res = []
for cfgno, dtype, nsub in ((3, 'float64', None), (4, 'float32', 4096), (2, 'float64', None)):
    x, y, cfg = synth.make_config(cfgno)
    if nsub:
        x, y = x[:nsub], y[:, :nsub]
    m = LCGP(y=y, x=x, q=cfg['q'], dtype=dtype)
    u = synth.param_points(cfgno, m._get_flat())[1]
    v, g = m.loss_and_grad(u)
    res += [np.array([v]), g]
np.save(sys.argv[2], np.concatenate(res))
