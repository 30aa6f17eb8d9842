"""Device-side state of one rank's share of the hot path: resident inputs, workspace, theta/out blocks.

PyTorch is plumbing here (device memory, the current HIP stream, pinned staging); all arithmetic of the path
runs in liblcgp_hip.so through the C ABI (include/lcgp_hip.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _hip

_DT = {"float64": _hip.F64, "f64": _hip.F64, "float32": _hip.F32, "f32": _hip.F32}


class HotPathEngine:
    """Holds x (n,d), Y (p,n), optional sr (n) and the workspace for `q_local` components on one GPU."""

    def __init__(self, x, Y, sr=None, q_local=1, dtype="float64", device=None):
        import torch
        _hip.require_gpu()
        self.lib = _hip.load()
        self.torch = torch
        self.dtype_name = "float64" if _DT[dtype] == _hip.F64 else "float32"
        self.dtype = _DT[dtype]
        self.tdtype = torch.float64 if self.dtype == _hip.F64 else torch.float32
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        x = np.ascontiguousarray(x, dtype=np.float64)
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        self.n, self.d = x.shape
        self.p = Y.shape[0]
        assert Y.shape[1] == self.n
        self.q_local = int(q_local)
        with torch.cuda.device(self.device):
            self.x = torch.as_tensor(x).to(self.device, self.tdtype).contiguous()
            self.Y = torch.as_tensor(Y).to(self.device, self.tdtype).contiguous()
            self.sr = None if sr is None else torch.as_tensor(np.ascontiguousarray(sr, np.float64)).to(
                self.device, self.tdtype).contiguous()
            nbytes = C.c_size_t(0)
            _hip.check(self.lib.lcgp_workspace_bytes(self.dtype, self.n, self.d, self.p, self.q_local, C.byref(nbytes)),
                       "lcgp_workspace_bytes")
            self.workspace_bytes = int(nbytes.value)
            self.workspace = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=self.device)
            self.tw = self.lib.lcgp_theta_width(self.d, self.p)
            self.ow = self.lib.lcgp_out_width(self.d, self.p)
            self.theta_dev = torch.zeros((self.q_local, self.tw), dtype=torch.float64, device=self.device)
            self.theta_pin = torch.zeros((self.q_local, self.tw), dtype=torch.float64).pin_memory()
            self.out_dev = torch.zeros((self.q_local, self.ow), dtype=torch.float64, device=self.device)
        self._theta_last = None

    # ------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _p(self, t):
        return C.c_void_p(0 if t is None else t.data_ptr())

    def upload_theta(self, theta_rows):
        theta_rows = np.asarray(theta_rows, dtype=np.float64).reshape(self.q_local, self.tw)
        self.theta_pin.copy_(self.torch.from_numpy(theta_rows))
        self.theta_dev.copy_(self.theta_pin, non_blocking=True)
        self._theta_last = theta_rows.copy()

    def enqueue(self):
        """One pass of the hot path over the resident theta block (asynchronous)."""
        with self.torch.cuda.device(self.device):
            _hip.check(self.lib.lcgp_nll_grad(self._stream(), self.dtype, self.n, self.d, self.p, self.q_local,
                                              self._p(self.x), self._p(self.Y), self._p(self.sr),
                                              self._p(self.theta_dev), self._p(self.workspace), self._p(self.out_dev)),
                       "lcgp_nll_grad")

    def evaluate(self, theta_rows):
        """theta rows (q_local, d+3+p) -> output rows (q_local, d+5+p) on the host (synchronises)."""
        self.upload_theta(theta_rows)
        self.enqueue()
        return self.out_dev.cpu().numpy()

    def evaluate_device(self, theta_rows):
        """Same, leaving the output block on the device (for an RCCL reduction before the D2H copy)."""
        self.upload_theta(theta_rows)
        self.enqueue()
        return self.out_dev

    def is_current(self, theta_rows):
        return self._theta_last is not None and np.array_equal(
            self._theta_last, np.asarray(theta_rows, np.float64).reshape(self.q_local, self.tw))

    # ------------------------------------------------------------------------------------------------
    def predict(self, x0s, same=False):
        """ghat, gvar (q_local, n0) for standardised x0s, from the factorisation of the last evaluate()."""
        torch = self.torch
        if self._theta_last is None:
            raise RuntimeError("predict() needs a preceding evaluate() at the current parameters")
        x0s = np.ascontiguousarray(x0s, np.float64)
        n0 = x0s.shape[0]
        assert x0s.shape[1] == self.d
        npad = (self.n + 127) // 128 * 128
        n0pad = (n0 + 63) // 64 * 64
        with torch.cuda.device(self.device):
            x0d = torch.as_tensor(x0s).to(self.device, self.tdtype).contiguous()
            scratch = torch.empty(2 * n0pad * npad, dtype=self.tdtype, device=self.device)
            ghat = torch.empty((self.q_local, n0), dtype=torch.float64, device=self.device)
            gvar = torch.empty((self.q_local, n0), dtype=torch.float64, device=self.device)
            _hip.check(self.lib.lcgp_predict(self._stream(), self.dtype, self.n, self.d, self.p, self.q_local,
                                             self._p(self.x), self._p(self.sr), self._p(self.theta_dev),
                                             self._p(self.workspace), n0, self._p(x0d), int(bool(same)),
                                             self._p(scratch), self._p(ghat), self._p(gvar)), "lcgp_predict")
            return ghat.cpu().numpy(), gvar.cpu().numpy()

    def fetch_vector(self, which, k):
        torch = self.torch
        with torch.cuda.device(self.device):
            out = torch.empty(self.n, dtype=self.tdtype, device=self.device)
            _hip.check(self.lib.lcgp_fetch_vector(self._stream(), self.dtype, self.n, self.d, self.p, self.q_local,
                                                  self._p(self.workspace), int(which), int(k), self._p(out)),
                       "lcgp_fetch_vector")
            return out.cpu().numpy().astype(np.float64)

    def fetch_matrix(self, which, k):
        torch = self.torch
        with torch.cuda.device(self.device):
            out = torch.empty((self.n, self.n), dtype=self.tdtype, device=self.device)
            _hip.check(self.lib.lcgp_fetch_matrix(self._stream(), self.dtype, self.n, self.d, self.p, self.q_local,
                                                  self._p(self.workspace), int(which), int(k), self._p(out)),
                       "lcgp_fetch_matrix")
            return out.cpu().numpy().astype(np.float64)


def matern32_device(x1, x2, ell, scale, nug, same, dtype="float64"):
    """covmat.py:31-55 on the GPU: returns the (n1, n2) matrix as a numpy float64 array."""
    import torch
    _hip.require_gpu()
    lib = _hip.load()
    dt = _DT[dtype]
    tdt = torch.float64 if dt == _hip.F64 else torch.float32
    dev = torch.device("cuda", torch.cuda.current_device())
    a = torch.as_tensor(np.ascontiguousarray(x1, np.float64)).to(dev, tdt).contiguous()
    b = torch.as_tensor(np.ascontiguousarray(x2, np.float64)).to(dev, tdt).contiguous()
    n1, d = a.shape
    n2 = b.shape[0]
    out = torch.empty((n1, n2), dtype=tdt, device=dev)
    ell = np.ascontiguousarray(ell, np.float64)
    _hip.check(lib.lcgp_matern32(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), dt, n1, n2, d,
                                 C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()),
                                 ell.ctypes.data_as(C.POINTER(C.c_double)), float(scale), float(nug), int(bool(same)),
                                 C.c_void_p(out.data_ptr())), "lcgp_matern32")
    return out.cpu().numpy().astype(np.float64)
