"""Where the time of ONE chain step goes: builds an instrumented copy of the library (build/libv_stamp.so: the product
source with s_memrealtime stamps in the special workgroup of chain_step_kernel and in the diagonal-block routine) and
prints the phases of the last chain step of an evaluation (n=4096, q=8: a late panel, no filler -- the latency floor
of the chain).  Experiment tool: nothing in the product reads the stamps or links the instrumented copy.

    python tools/phase_stamps.py build       # here (hipcc, no GPU needed)
    python tools/phase_stamps.py run         # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'lcgp_amd', 'csrc', 'lcgp_hip.hip')
OUT = os.path.join(ROOT, 'build', 'libv_stamp.so')

# (anchor in the product source, replacement); every anchor must occur exactly once
PATCHES = [
    ("__device__ __forceinline__ double fast_rcp(double a) {",
     "__device__ unsigned long long g_st[4][40];\n"
     "#define STAMP(slot) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_st[threadIdx.x >> 6][slot] = "
     "__builtin_amdgcn_s_memrealtime(); } while (0)\n"
     "__device__ __forceinline__ double fast_rcp(double a) {"),
    ("    double (*S)[17] = (double (*)[17])scratch;\n    int first_bad = 0;",
     "    double (*S)[17] = (double (*)[17])scratch;\n    int first_bad = 0;\n"
     "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    STAMP(5);"),
    ("            double dmine = 1.0;                       // the pivot of row li of the diagonal block\n",
     "            STAMP(22 + 4 * kb);\n            double dmine = 1.0;                       // the pivot of row li of the diagonal block\n"),
    ("            const double rsl = fast_rsqrt(dmine);     // off the chain\n",
     "            STAMP(23 + 4 * kb);\n            const double rsl = fast_rsqrt(dmine);     // off the chain\n"),
    ("            if (lq == 0) { dinv[base + li] = rsl; pivs[base + li] = dmine; }\n",
     "            STAMP(24 + 4 * kb);\n            if (lq == 0) { dinv[base + li] = rsl; pivs[base + li] = dmine; }\n"),
    ("            if (lane == 0) bad[kb] = first_bad;\n        } else if (wv == kb - 1) {",
     "            if (lane == 0) bad[kb] = first_bad;\n            STAMP(6 + 3 * kb);\n        } else if (wv == kb - 1) {"),
    ("        __syncthreads();\n        if (wv > kb) {", "        __syncthreads();\n        STAMP(7 + 3 * kb);\n        if (wv > kb) {"),
    ("                                                                       0, 0, 0);\n            }\n        }\n    }\n"
     "    // row 3 of the inverse",
     "                                                                       0, 0, 0);\n            }\n        }\n"
     "        STAMP(8 + 3 * kb);\n    }\n    STAMP(18);\n    // row 3 of the inverse"),
    ("    else leaf_store_l_panel<T>(Mb, npad, lt, 3, lane);\n    __syncthreads();\n",
     "    else leaf_store_l_panel<T>(Mb, npad, lt, 3, lane);\n    __syncthreads();\n    STAMP(19);\n"),
    ("    leaf_factor_invert<T, FROM_LDS>(Mb, Wb, npad, lt, w, scratch, dinv, pivs, bad, jb);",
     "    STAMP(4);\n    leaf_factor_invert<T, FROM_LDS>(Mb, Wb, npad, lt, w, scratch, dinv, pivs, bad, jb);\n    STAMP(20);"),
    ("            if (fb && info_prev == 0) info[k] = fb;\n        }\n    }\n}",
     "            if (fb && info_prev == 0) info[k] = fb;\n        }\n    }\n    STAMP(21);\n}"),
    ("        T pw[TL::SPT][TL::EPT];\n", "        T pw[TL::SPT][TL::EPT];\n        STAMP(0);\n"),
    ("        TL::to_operand(acc, F, lane, wm0, wn0);\n        TL::zero(acc);",
     "        STAMP(1);\n        TL::to_operand(acc, F, lane, wm0, wn0);\n        TL::zero(acc);"),
    ("        TL::store(acc, Ct, ld, lane, wm0, wn0);         // L[r, c]\n",
     "        TL::store(acc, Ct, ld, lane, wm0, wn0);         // L[r, c]\n        STAMP(2);\n"),
    ("            TL::template mma_ab_lds<true>(dacc, F, lane, wm0, wn0);\n",
     "            TL::template mma_ab_lds<true>(dacc, F, lane, wm0, wn0);\n            STAMP(3);\n"),
    ("const char* lcgp_source_hash(void) { return LCGP_SRC_HASH; }",
     "const char* lcgp_source_hash(void) { return LCGP_SRC_HASH; }\n"
     "int lcgp_debug_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_st), sizeof(g_st)); }"),
]

NAMES = {0: 'start', 1: 'previous column applied', 2: 'L[c+1,c] = tile W_cc^T stored', 3: 'diagonal block updated',
         4: 'diagonal-block routine entered', 5: 'block in registers (from LDS)', 18: 'panels done', 19: 'row 3 of the inverse',
         20: 'last rows of L and W issued', 21: 'log-determinant stored'}
for _kb in range(4):
    NAMES[22 + 4 * _kb] = 'panel %d: in chain layout' % _kb
    NAMES[23 + 4 * _kb] = 'panel %d: 16 pivots done' % _kb
    NAMES[24 + 4 * _kb] = 'panel %d: scaled' % _kb
    NAMES[6 + 3 * _kb] = 'panel %d factored (wave %d)' % (_kb, _kb)
    NAMES[7 + 3 * _kb] = 'panel %d barrier' % _kb
    NAMES[8 + 3 * _kb] = 'panel %d applied' % _kb


def build():
    s = open(SRC).read().replace('#include "../../include/lcgp_hip.h"', '#include "%s"' % os.path.join(ROOT, 'include', 'lcgp_hip.h'))
    s = s.replace('#include "fill_sched.h"', '#include "%s"' % os.path.join(ROOT, 'lcgp_amd', 'csrc', 'fill_sched.h'))
    for old, new in PATCHES:
        assert s.count(old) == 1, 'anchor drifted: %r' % old[:60]
        s = s.replace(old, new)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    tmp = os.path.join(ROOT, 'build', 'v_stamp.hip')
    open(tmp, 'w').write(s)
    subprocess.check_call(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-o', OUT, tmp])
    print('built', OUT)


def run():
    import numpy as np
    sys.path.insert(0, ROOT)
    from lcgp_amd import _hip
    _hip.LIB_PATH = OUT
    from lcgp_amd import LCGP, synth
    import torch
    x, y, cfg = synth.make_config(3)
    m = LCGP(y=y, x=x, q=8)
    u = m._get_flat()
    for _ in range(3):
        m.loss_and_grad(u)
    torch.cuda.synchronize()
    lib = _hip.load()
    out = np.zeros((4, 40), np.uint64)
    lib.lcgp_debug_stamps.restype = C.c_int
    lib.lcgp_debug_stamps.argtypes = [C.c_void_p]
    assert lib.lcgp_debug_stamps(out.ctypes.data) == 0
    t0 = int(out[0, 0])
    print('# us since the special workgroup of the last chain step started (100 MHz counter), one column per wave')
    for s in range(38):
        row = ['%8.2f' % ((int(out[w, s]) - t0) / 100.0) if out[w, s] else '       -' for w in range(4)]
        print('%2d %-40s %s' % (s, NAMES.get(s, ''), ' '.join(row)))


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1] if len(sys.argv) > 1 else 'build']()
