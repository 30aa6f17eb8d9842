"""ctypes binding of liblcgp_hip.so (the C ABI declared in include/lcgp_hip.h).

The library is built in-tree by `make` / `__graft_entry__.build()`.  There is NO CPU fallback: if the
shared object is missing or no GPU is visible, every hot-path call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# (LCGP_HIP_LIB: tools only -- e.g. the stamped build of `make trace`; tests and the package use the in-tree library)
LIB_PATH = os.environ.get("LCGP_HIP_LIB") or os.path.join(_HERE, "liblcgp_hip.so")
SRC_PATH = os.path.join(_HERE, "csrc", "lcgp_hip.hip")
HDR_PATH = os.path.join(_ROOT, "include", "lcgp_hip.h")
SCHED_PATH = os.path.join(_HERE, "csrc", "fill_sched.h")

F64, F32 = 0, 1
KERNELS = {"matern32": 0, "se": 1}      # include/lcgp_hip.h: LCGP_KERNEL_MATERN32 / LCGP_KERNEL_SE (an extension, parity unpinned)


class Sched(C.Structure):
    """lcgp_sched of include/lcgp_hip.h: launch shapes of the factorisation / inverse, passed per call."""
    _fields_ = [("outer_blocks", C.c_int), ("syrk_small_tiles", C.c_int), ("trtri_small_tiles", C.c_int),
                ("lauum_small_tiles", C.c_int), ("trtri_level_small", C.c_int), ("fill_leaf", C.c_int),
                ("fill_step", C.c_int), ("leaf_in_wide", C.c_int), ("progressive_tiles", C.c_int), ("progressive_far", C.c_int), ("progressive_lauum", C.c_int), ("pair_tiles", C.c_int)]


# every symbol include/lcgp_hip.h declares: name -> (restype, argtypes)
_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
_sp = C.POINTER(Sched)
SIGNATURES = {
    "lcgp_version": (_i, []),
    "lcgp_source_hash": (C.c_char_p, []),
    "lcgp_last_error": (C.c_char_p, []),
    "lcgp_theta_width": (_i, [_i, _i]),
    "lcgp_out_width": (_i, [_i, _i]),
    "lcgp_partial_width": (_i, [_i, _i, _i]),
    "lcgp_sched_default": (_i, [_sp]),
    "lcgp_workspace_bytes": (_i, [_i, _i, _i, _i, _i, C.POINTER(C.c_size_t)]),
    "lcgp_predict_scratch_bytes": (_i, [_i, _i, _i, _i, C.POINTER(C.c_size_t)]),
    "lcgp_matern32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, C.POINTER(_d), _d, _d, _i, _vp]),
    "lcgp_covmat": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, C.POINTER(_d), _d, _d, _i, _vp]),
    "lcgp_kernel_build": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "lcgp_potrf_logdet": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sp, _vp]),
    "lcgp_potri": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _sp]),
    "lcgp_trtri": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _sp]),
    "lcgp_lauum": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _sp]),
    "lcgp_lauum_clock": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lcgp_fetch_matrix": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "lcgp_fetch_vector": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "lcgp_nll_grad": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sp, _vp]),
    "lcgp_plan_bytes": (_i, [_i, _i, _i, _i, _sp, C.POINTER(C.c_size_t)]),
    "lcgp_plan_build": (_i, [_i, _i, _i, _i, _sp, _vp, C.c_size_t]),
    "lcgp_plan_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "lcgp_pack_partial": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lcgp_predict": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i]),
}

_lib = None


def source_hash() -> str:
    """sha256 over the sources the library is built from (first 16 hex digits); compiled into the binary as
    LCGP_SRC_HASH and returned by lcgp_source_hash(), so source and binary can be compared on any box."""
    import hashlib
    h = hashlib.sha256()
    for path in (SRC_PATH, HDR_PATH, SCHED_PATH):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def binary_hash():
    """LCGP_SRC_HASH of the shared object on disk (read from the file, the library is not loaded), or None."""
    import re
    if not os.path.exists(LIB_PATH):
        return None
    with open(LIB_PATH, "rb") as f:
        m = re.search(rb"LCGP_SRC_HASH=([0-9a-f]{16})", f.read())
    return m.group(1).decode() if m else None


def needs_build() -> bool:
    return binary_hash() != source_hash()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> lcgp_amd/liblcgp_hip.so (cross-compiles without a GPU).  Rebuilds whenever the
    hash embedded in the binary differs from the hash of the sources (not an mtime test)."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           '-DLCGP_SRC_HASH="LCGP_SRC_HASH=%s"' % source_hash(), "-o", LIB_PATH, SRC_PATH]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building liblcgp_hip.so failed:\n" + res.stderr)
    return LIB_PATH


def load():
    """Returns the loaded library with argtypes set; raises if it is not there (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "lcgp_amd: %s is missing -- build it with `make` or `python -c 'import __graft_entry__ as g; g.build()'`. "
            "The LCGP hot path has no CPU fallback." % LIB_PATH)
    # torch must own the HIP runtime of the process (same SONAME, loaded first)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def loaded_hash() -> str:
    """The source hash the LOADED library reports."""
    return load().lcgp_source_hash().decode().split("=")[-1]


def default_sched() -> Sched:
    s = Sched()
    check(load().lcgp_sched_default(C.byref(s)), "lcgp_sched_default")
    return s


def check(rc: int, what: str):
    if rc != 0:
        msg = load().lcgp_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("lcgp_amd: the LCGP hot path runs on an AMD GPU (gfx950) only; no GPU is visible "
                           "and there is deliberately no CPU fallback.")
