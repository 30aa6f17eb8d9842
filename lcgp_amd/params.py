"""Parameter container + SoftClip transforms of the hot path (host side, P scalars).

Mirrors what the reference gets from `gpflow.Parameter(..., transform=tfp.bijectors.SoftClip(low, high))`
(lcgp.py:181-211): the optimiser sees the UNCONSTRAINED value, every read gives the CONSTRAINED one.
SoftClip with hinge softness 1:
    v = hi - (hi-lo)/softplus(hi-lo) * softplus(hi - lo - softplus(u - lo))
"""
from __future__ import annotations

import numpy as np

F64 = np.float64


def _softplus(t):
    return np.logaddexp(0.0, t)


def _softplus_inv(t):
    return t + np.log(-np.expm1(-t))


def _sigmoid(t):
    return 0.5 * (1.0 + np.tanh(0.5 * t))


class SoftClip:
    def __init__(self, low: float, high: float):
        self.low = float(low)
        self.high = float(high)
        self._w = self.high - self.low
        self._c = self._w / float(_softplus(self._w))

    def forward(self, u):
        u = np.asarray(u, F64)
        return self.high - self._c * _softplus(self._w - _softplus(u - self.low))

    def inverse(self, v):
        v = np.asarray(v, F64)
        return self.low + _softplus_inv(self._w - _softplus_inv((self.high - v) / self._c))

    def dforward(self, u):
        u = np.asarray(u, F64)
        return self._c * _sigmoid(self._w - _softplus(u - self.low)) * _sigmoid(u - self.low)


def softclip_flat(u, lo, hi, w, c):
    """SoftClip.forward and SoftClip.dforward of a FLAT vector whose elements have their own bounds (arrays lo, hi, w = hi - lo,
    c = w / softplus(w)): one pass for the three bounded parameter blocks of an evaluation, the inner softplus shared.  Element
    by element the same operations as the two methods above."""
    t = u - lo
    a = w - _softplus(t)
    return hi - c * _softplus(a), c * _sigmoid(a) * _sigmoid(t)


class Identity:
    low = -np.inf
    high = np.inf

    def forward(self, u):
        return np.asarray(u, F64)

    def inverse(self, v):
        return np.asarray(v, F64)

    def dforward(self, u):
        return np.ones_like(np.asarray(u, F64))


class _Variable:
    """What `trainable_variables` yields: the unconstrained array, with `.numpy()` and `.name`."""

    def __init__(self, param):
        self._p = param
        self.name = param.name + ":0"

    def numpy(self):
        return self._p.unconstrained.copy()

    @property
    def shape(self):
        return self._p.unconstrained.shape


class Parameter:
    """Constrained-on-read parameter (gpflow.Parameter surface used by the reference and its tests)."""

    def __init__(self, value, name: str, transform=None):
        self.name = name
        self.transform = transform if transform is not None else Identity()
        self.unconstrained = np.array(self.transform.inverse(np.asarray(value, F64)), dtype=F64)

    # -- constrained view ---------------------------------------------------------------------------
    def numpy(self):
        return np.array(self.transform.forward(self.unconstrained), dtype=F64)

    def assign(self, value):
        value = np.asarray(value, F64)
        if value.shape != self.unconstrained.shape:
            raise ValueError("shape mismatch in assign: %s vs %s" % (value.shape, self.unconstrained.shape))
        self.unconstrained = np.array(self.transform.inverse(value), dtype=F64)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, idx):
        import torch
        return torch.as_tensor(self.numpy()[idx])

    @property
    def shape(self):
        return self.unconstrained.shape

    @property
    def size(self):
        return self.unconstrained.size

    def variable(self):
        return _Variable(self)

    def __repr__(self):
        return "Parameter(%s, shape=%s, value=%s)" % (self.name, self.shape, np.array2string(self.numpy(), precision=5))
