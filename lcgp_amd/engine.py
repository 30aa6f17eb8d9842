"""Device-side state of one rank's share of the hot path: resident inputs, workspace, theta/out blocks.

PyTorch is plumbing here (device memory, the current HIP stream, pinned staging); all arithmetic of the path
runs in liblcgp_hip.so through the C ABI (include/lcgp_hip.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _hip

_DT = {"float64": _hip.F64, "f64": _hip.F64, "float32": _hip.F32, "f32": _hip.F32}

# rows of x0 handled per lcgp_predict call: bounds the scratch (2 * q_local * chunk * npad elements) however many
# new inputs a caller passes (the reference has no limit on n0 either)
PREDICT_CHUNK = 2048


class HotPathEngine:
    """Holds x (n,d), Y (p,n), optional sr (n) and the workspace for the local components on one GPU.

    `comp_ids` are the GLOBAL indices of the local components (k -> rank k mod G) and `q_total` the number of
    components over all ranks; they place this rank's gradient slots in the vector the ranks all-reduce
    (`evaluate_partial`).  Defaults: a single rank holding components 0 .. q_local-1."""

    def __init__(self, x, Y, sr=None, q_local=1, dtype="float64", device=None, comp_ids=None, q_total=None, kernel="matern32"):
        import torch
        _hip.require_gpu()
        self.lib = _hip.load()
        self.kernel_id = _hip.KERNELS[kernel]      # covariance kernel of the latent components (the reference: Matern-3/2 only)
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
        comp_ids = list(range(self.q_local)) if comp_ids is None else [int(k) for k in comp_ids]
        assert len(comp_ids) == self.q_local
        self.q_total = int(q_total) if q_total is not None else (max(comp_ids) + 1 if comp_ids else 1)
        assert all(0 <= k < self.q_total for k in comp_ids)
        self._sched_obj = None       # an _hip.Sched to override the launch schedule (tests / tools); None = defaults
        self.use_plan = True         # the launch plan is built once per schedule (lcgp_plan_build) and passed with every call
        self._plan_cache = {}        # with_inverse -> host block
        with torch.cuda.device(self.device):
            self.x = torch.as_tensor(x).to(self.device, self.tdtype).contiguous()
            self.Y = torch.as_tensor(Y).to(self.device, self.tdtype).contiguous()
            self.sr = None if sr is None else torch.as_tensor(np.ascontiguousarray(sr, np.float64)).to(
                self.device, self.tdtype).contiguous()
            nbytes = C.c_size_t(0)
            _hip.check(self.lib.lcgp_workspace_bytes(self.dtype, self.n, self.d, self.p, self.q_local, C.byref(nbytes)),
                       "lcgp_workspace_bytes")
            self.workspace_bytes = int(nbytes.value)
            # zero-filled, i.e. touched once here: the first evaluation of a fresh engine otherwise pays 30-70 ms of first-touch
            # page mapping for its 3 x q_local matrices inside the optimiser's first step (and the clock words start at zero)
            self.workspace = torch.zeros(self.workspace_bytes, dtype=torch.uint8, device=self.device)
            self.tw = self.lib.lcgp_theta_width(self.d, self.p)
            self.ow = self.lib.lcgp_out_width(self.d, self.p)
            self.pw = self.lib.lcgp_partial_width(self.d, self.p, self.q_total)
            # one upload per evaluation: the theta rows and, behind them, the guard word of the lock-step check
            self._theta_flat = torch.zeros(self.q_local * self.tw + 1, dtype=torch.float64, device=self.device)
            self.theta_dev = self._theta_flat[:self.q_local * self.tw].view(self.q_local, self.tw)
            self.guard_dev = self._theta_flat[self.q_local * self.tw:]
            # two pinned staging rows used alternately: the H2D copy of one evaluation may still be in flight when the
            # host packs the next one (evaluate() itself synchronises, enqueue-style callers do not)
            self._theta_pin = [torch.zeros(self.q_local * self.tw + 1, dtype=torch.float64).pin_memory() for _ in range(2)]
            self._theta_pin_np = [t.numpy() for t in self._theta_pin]      # the same memory, for host-side writes
            self._pin_event = [None, None]
            self._pin_next = 0
            self._nll_ptrs = None
            self._pack_ptrs = None
            self.out_dev = torch.zeros((self.q_local, self.ow), dtype=torch.float64, device=self.device)
            self.comp_dev = torch.as_tensor(np.asarray(comp_ids, np.int32)).to(self.device)
            self.partial_dev = torch.zeros(self.pw, dtype=torch.float64, device=self.device)
            self._scratch = None
        self._theta_last = None

    # ------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _p(self, t):
        return C.c_void_p(0 if t is None else t.data_ptr())

    @property
    def sched(self):
        return self._sched_obj

    @sched.setter
    def sched(self, s):
        self._sched_obj = s
        self._plan_cache = {}        # a plan carries the schedule it was built for

    def _sched(self):
        return None if self._sched_obj is None else C.byref(self._sched_obj)

    def _build_plan(self, with_inverse):
        """the launch plan of the factorisation for the current schedule as a numpy byte block (lcgp_plan_build): host-only,
        a function of (dtype, n, q_local, with_inverse, schedule) and nothing else"""
        key = bool(with_inverse)
        nbytes = C.c_size_t(0)
        _hip.check(self.lib.lcgp_plan_bytes(self.dtype, self.n, self.q_local, int(key), self._sched(), C.byref(nbytes)),
                   "lcgp_plan_bytes")
        host = np.zeros(int(nbytes.value), dtype=np.uint8)
        _hip.check(self.lib.lcgp_plan_build(self.dtype, self.n, self.q_local, int(key), self._sched(),
                                            C.c_void_p(host.ctypes.data), nbytes), "lcgp_plan_build")
        return host

    def plan(self, with_inverse=True):
        """host pointer of the launch plan of the factorisation for the current schedule: planned ONCE (lcgp_plan_build), the
        position-independent block kept in host memory and passed with every evaluation.  NULL with `use_plan = False`:
        the library then plans per call."""
        if not self.use_plan:
            return C.c_void_p(0)
        key = bool(with_inverse)
        if key not in self._plan_cache:
            self._plan_cache[key] = self._build_plan(key)
        return C.c_void_p(self._plan_cache[key].ctypes.data)

    def plan_info(self, with_inverse=True):
        """launches / what the plan leaves behind the factorisation (lcgp_plan_info)"""
        key = bool(with_inverse)
        host = self._plan_cache.get(key) if self.use_plan else None
        if host is None:
            host = self._build_plan(key)        # (use_plan = False: a temporary plan, what the library would plan per call)
            if self.use_plan:
                self._plan_cache[key] = host
        v = [C.c_int(0) for _ in range(2)]
        _hip.check(self.lib.lcgp_plan_info(C.c_void_p(host.ctypes.data), *[C.byref(x) for x in v]), "lcgp_plan_info")
        return dict(zip(("launches", "inverse_done"), (x.value for x in v)))

    def upload_theta(self, theta_rows, guard=0.0, stream=None):
        torch = self.torch
        theta_rows = np.asarray(theta_rows, dtype=np.float64).reshape(self.q_local, self.tw)
        i = self._pin_next
        self._pin_next ^= 1
        ev = self._pin_event[i]
        if ev is not None:
            ev.synchronize()                      # the copy that last read this staging buffer has completed
        else:
            ev = self._pin_event[i] = torch.cuda.Event()
        # (the pinned staging rows are written through their numpy views: no torch op per evaluation on the host side)
        pin = self._theta_pin_np[i]
        pin[:-1] = theta_rows.reshape(-1)
        pin[-1] = float(guard)
        if stream is None:
            with torch.cuda.device(self.device):
                self._theta_flat.copy_(self._theta_pin[i], non_blocking=True)
                ev.record(torch.cuda.current_stream(self.device))
        else:                                     # (the caller is inside the device context and holds the current stream)
            self._theta_flat.copy_(self._theta_pin[i], non_blocking=True)
            ev.record(stream)
        self._theta_last = theta_rows.copy()

    def _nll_args(self):
        """the constant pointer arguments of lcgp_nll_grad as ctypes objects, made once (the tensors live as long as the engine)"""
        if self._nll_ptrs is None:
            self._nll_ptrs = tuple(self._p(t) for t in (self.x, self.Y, self.sr, self.theta_dev, self.workspace, self.out_dev))
        return self._nll_ptrs

    def enqueue(self, stream=None):
        """One pass of the hot path over the resident theta block (asynchronous)."""
        if stream is not None:                    # (inside the device context already)
            _hip.check(self.lib.lcgp_nll_grad(C.c_void_p(stream.cuda_stream), self.dtype, self.kernel_id, self.n, self.d, self.p, self.q_local,
                                              *self._nll_args(), self._sched(), self.plan(True)), "lcgp_nll_grad")
            return
        with self.torch.cuda.device(self.device):
            _hip.check(self.lib.lcgp_nll_grad(self._stream(), self.dtype, self.kernel_id, self.n, self.d, self.p, self.q_local,
                                              *self._nll_args(), self._sched(), self.plan(True)),
                       "lcgp_nll_grad")

    def evaluate(self, theta_rows):
        """theta rows (q_local, d+3+p) -> output rows (q_local, d+5+p) on the host (synchronises)."""
        self.upload_theta(theta_rows)
        self.enqueue()
        return self.out_dev.cpu().numpy()

    def evaluate_partial(self, theta_rows, guard=0.0):
        """theta rows -> this rank's share of the reduced vector, LEFT ON THE DEVICE (lcgp_pack_partial): the caller
        all-reduces it in place over the ranks (RCCL) and copies it to the host once.  `guard` travels in its last slot."""
        if self._pack_ptrs is None:
            self._pack_ptrs = tuple(self._p(t) for t in (self.comp_dev, self.theta_dev, self.out_dev, self.guard_dev, self.partial_dev))
        torch = self.torch
        with torch.cuda.device(self.device):      # one device context and one stream lookup per evaluation
            st = torch.cuda.current_stream(self.device)
            self.upload_theta(theta_rows, guard, st)
            self.enqueue(st)
            _hip.check(self.lib.lcgp_pack_partial(C.c_void_p(st.cuda_stream), self.d, self.p, self.q_local, self.q_total,
                                                  *self._pack_ptrs), "lcgp_pack_partial")
        return self.partial_dev

    def is_current(self, theta_rows):
        return self._theta_last is not None and np.array_equal(
            self._theta_last, np.asarray(theta_rows, np.float64).reshape(self.q_local, self.tw))

    # ------------------------------------------------------------------------------------------------
    def predict_block(self, x0s, same=False):
        """(2, q_local, n0) float64 DEVICE tensor [ghat; gvar] for standardised x0s, from the factorisation of the last
        evaluate().  x0 is processed in chunks of PREDICT_CHUNK rows with one engine-owned scratch buffer."""
        torch = self.torch
        if self._theta_last is None:
            raise RuntimeError("predict() needs a preceding evaluate() at the current parameters")
        x0s = np.ascontiguousarray(x0s, np.float64)
        n0 = x0s.shape[0]
        assert x0s.shape[1] == self.d
        chunk = min(n0, PREDICT_CHUNK)
        with torch.cuda.device(self.device):
            x0d = torch.as_tensor(x0s).to(self.device, self.tdtype).contiguous()
            nbytes = C.c_size_t(0)
            _hip.check(self.lib.lcgp_predict_scratch_bytes(self.dtype, self.n, self.q_local, chunk, C.byref(nbytes)),
                       "lcgp_predict_scratch_bytes")
            if self._scratch is None or self._scratch.numel() < nbytes.value:
                self._scratch = None
                self._scratch = torch.empty(int(nbytes.value), dtype=torch.uint8, device=self.device)
            # ONE (2, q_local, n0) result block; every chunk writes its columns in place (row stride = n0): no per-chunk
            # temporaries, no device-to-device copies
            out = torch.empty((2, self.q_local, n0), dtype=torch.float64, device=self.device)
            ghat, gvar = out[0], out[1]
            st, xp, srp, thp, wsp, scp = self._stream(), self._p(self.x), self._p(self.sr), self._p(self.theta_dev), \
                self._p(self.workspace), self._p(self._scratch)
            for lo in range(0, n0, chunk):
                m = min(chunk, n0 - lo)
                # the nugget term only exists when x0 IS the training set (covmat.py:46-51): then n0 == n and the
                # diagonal of the full cross matrix falls on rows lo .. lo+m of this chunk
                _hip.check(self.lib.lcgp_predict(st, self.dtype, self.kernel_id, self.n, self.d, self.p, self.q_local, xp, srp, thp, wsp, m,
                                                 C.c_void_p(x0d.data_ptr() + lo * self.d * x0d.element_size()),
                                                 (1 + lo) if same else 0, scp,
                                                 C.c_void_p(ghat.data_ptr() + 8 * lo), C.c_void_p(gvar.data_ptr() + 8 * lo), n0),
                           "lcgp_predict")
            return out

    def predict_device(self, x0s, same=False):
        """ghat, gvar (q_local, n0): the two halves of predict_block()"""
        out = self.predict_block(x0s, same)
        return out[0], out[1]

    def predict(self, x0s, same=False):
        ghat, gvar = self.predict_device(x0s, same)
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


def matern32_device(x1, x2, ell, scale, nug, same, dtype="float64", kernel="matern32"):
    """covmat.py:31-55 on the GPU: returns the (n1, n2) matrix as a numpy float64 array (kernel = "se": the squared-exponential
    product kernel with the same scale / nugget structure, an extension the reference does not have)."""
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
    _hip.check(lib.lcgp_covmat(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), dt, _hip.KERNELS[kernel], n1, n2, d,
                                 C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()),
                                 ell.ctypes.data_as(C.POINTER(C.c_double)), float(scale), float(nug), int(bool(same)),
                                 C.c_void_p(out.data_ptr())), "lcgp_covmat")
    return out.cpu().numpy().astype(np.float64)
