"""Enqueues a few evaluations of the headline configuration without reading results (for kernel traces of builds whose
numbers are deliberately garbage: tools/build_variant.sh ... -DLCGP_EXP=n).

    python tools/run_enqueue.py <q> <evaluations>
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lcgp_amd import LCGP, synth  # noqa: E402

q, nev = int(sys.argv[1]), int(sys.argv[2])
x, y, cfg = synth.make_config(3)
m = LCGP(y=y, x=x, q=q, dtype='float64')
eng = m._get_engine()
sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
eng.upload_theta(m._theta_rows(sig_eff))
for _ in range(nev):
    eng.enqueue()
torch.cuda.synchronize()
print('done')
