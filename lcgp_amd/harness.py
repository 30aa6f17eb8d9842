"""Run harness with the call pattern of the reference's examples (docs/call_model.py:5-86 ==
illustration-examples/call_model.py): `LCGPRun(runno=, data=, ...)`, then `.define_model()`, `.train()`,
`.predict()`; plus the four summary functions that file defines beside it (call_model.py:89-126), whose
conventions differ from `lcgp.evaluation` (per-row range/std normalisation, z = 1.96, mean instead of sum).

    data = dict(xtrain=..., ytrain=..., xtest=..., ytest=...[, ytrue=..., ystd=...])   # y arrays are (p, n)
    run = LCGPRun(runno='r0', data=data, submethod='rep', num_latent=3)
    run.define_model(); run.train(); mean, predvar, confvar = run.predict()
"""
import numpy as np

from .lcgp import LCGP


class SuperRun:
    """Holds one train/test split (call_model.py:5-32)."""

    def __init__(self, runno, data, verbose=False, **kwargs):
        self.data = data
        for key in ('xtrain', 'ytrain', 'xtest', 'ytest'):
            setattr(self, key, data[key])
        for key in ('ytrue', 'ystd'):            # optional
            if key in data:
                setattr(self, key, data[key])
        self.runno = runno
        self.model = None
        self.modelname = ''
        self.n = self.xtrain.shape[0]
        self.num_output = self.ytrain.shape[0]
        self.verbose = verbose

    def define_model(self):
        pass

    def train(self):
        pass

    def predict(self):
        pass


class LCGPRun(SuperRun):
    """call_model.py:35-86.  Extra keyword-only pass-throughs of this build: device, dtype, process_group."""

    def __init__(self, submethod='full', robust=True, err_struct=None, num_latent=None, var_threshold=None,
                 device=None, dtype='float64', process_group=None, **kwargs):
        super().__init__(**kwargs)
        self.modelname = 'LCGP_robust' if robust else 'LCGP'
        self.num_latent = num_latent
        self.var_threshold = var_threshold
        self.submethod = submethod
        self.robust = robust
        self.err_struct = err_struct
        self._extra = dict(device=device, dtype=dtype, process_group=process_group)

    def define_model(self):
        self.model = LCGP(y=self.ytrain, x=self.xtrain, parameter_clamp_flag=False, q=self.num_latent,
                          var_threshold=self.var_threshold, diag_error_structure=self.err_struct,
                          robust_mean=self.robust, submethod=self.submethod, **self._extra)

    def train(self):
        self.model.fit(verbose=self.verbose)

    def predict(self, train=False, return_fullcov=False, as_pxn=False):
        """numpy (ymean, ypredvar, yconfvar[, fullcov]); `as_pxn` transposes the three (p, n0) arrays.  The replicated
        path has no full covariance (lcgp.py:929): its fourth item stays None (the reference calls .numpy() on it)."""
        out = self.model.predict(self.xtrain if train else self.xtest, return_fullcov=return_fullcov)
        arrays = [t.numpy() for t in out[:3]]
        if as_pxn:
            arrays = [a.T for a in arrays]
        if return_fullcov:
            arrays.append(None if out[3] is None else out[3].numpy())
        return tuple(arrays)


# ---- summaries defined next to the harness in the reference (call_model.py:89-126) -------------------------------
def rmse(ytrue, yhat):
    return float(np.sqrt(np.mean((ytrue - yhat) ** 2)))


def normalized_rmse(ytrue, yhat, method="range"):
    """Mean over the output rows of RMSE_row / (range or std of that row); zero spread counts as 1."""
    if method == "range":
        spread = np.ptp(ytrue, axis=1, keepdims=True)
    elif method == "std":
        spread = np.std(ytrue, axis=1, ddof=0, keepdims=True)
    else:
        raise ValueError("method must be 'range' or 'std'")
    spread = np.where(spread == 0, 1.0, spread)
    per_row = np.sqrt(np.mean((ytrue - yhat) ** 2, axis=1, keepdims=True)) / spread
    return float(np.mean(per_row))


def intervalstats(ytrue, mean, var, z=1.96):
    """Coverage and mean width of mean +- z sd over all outputs and points."""
    half = z * np.sqrt(var)
    inside = (ytrue >= mean - half) & (ytrue <= mean + half)
    return float(np.mean(inside)), float(np.mean(2 * half))


def dss(ytrue, mean, var, use_diag=True):
    """Mean over all entries of (y - mu)^2 / s2 + log s2, s2 floored at 1e-12."""
    s2 = np.maximum(var, 1e-12)
    return float(np.mean((ytrue - mean) ** 2 / s2 + np.log(s2)))
