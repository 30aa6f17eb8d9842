"""How long the chain workgroup of a hosted panel launch (lcgp_sched.hosted, host_kernel) takes, alone and beside the
deferred-update tiles of the same launch: builds an instrumented copy of the library (build/libv_hstamp.so: the product source
with s_memrealtime stamps around chain_panel_body and an atomic max of the end stamps of the hosted tiles) and prints, per
panel, the duration of the chain workgroup of component 0 and the time from its start to the end of the last hosted tile.
Experiment tool: nothing in the product reads the stamps or links the instrumented copy.

    python tools/host_stamps.py build                          # here (hipcc, no GPU needed)
    python tools/host_stamps.py run [q] [field=value ...]      # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'lcgp_amd', 'csrc', 'lcgp_hip.hip')
OUT = os.path.join(ROOT, 'build', 'libv_hstamp.so')

PATCHES = [
    ("template <typename T>\nconstexpr int host_lds_bytes() {",
     "__device__ unsigned long long g_hst[64][4];\n"
     "template <typename T>\nconstexpr int host_lds_bytes() {"),
    ("        if (threadIdx.x >= 256) return;\n        chain_panel_body<T>(lds, blockIdx.x, a);\n        return;",
     "        if (threadIdx.x >= 256) return;\n"
     "        if (blockIdx.x == 0 && threadIdx.x == 0) g_hst[a.J / 4][0] = __builtin_amdgcn_s_memrealtime();\n"
     "        chain_panel_body<T>(lds, blockIdx.x, a);\n"
     "        __syncthreads();\n"
     "        if (blockIdx.x == 0 && threadIdx.x == 0) g_hst[a.J / 4][1] = __builtin_amdgcn_s_memrealtime();\n"
     "        return;"),
    ("    host_tile_body<T>(a, blockIdx.x - a.q, lds);\n}",
     "    host_tile_body<T>(a, blockIdx.x - a.q, lds);\n"
     "    __syncthreads();\n"
     "    if (threadIdx.x == 0) { atomicMax(&g_hst[a.J / 4][2], (unsigned long long)__builtin_amdgcn_s_memrealtime());\n"
     "                            atomicAdd(&g_hst[a.J / 4][3], 1ull); }\n}"),
    ("const char* lcgp_source_hash(void) { return LCGP_SRC_HASH; }",
     "const char* lcgp_source_hash(void) { return LCGP_SRC_HASH; }\n"
     "int lcgp_debug_host_stamps(unsigned long long* out, int clear) {\n"
     "    if (clear) { static unsigned long long z[64][4]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_hst), z, sizeof(z)); }\n"
     "    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hst), sizeof(g_hst)); }"),
]


def build():
    src = open(SRC).read()
    for old, new in PATCHES:
        assert src.count(old) == 1, old
        src = src.replace(old, new)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    tmp = os.path.join(ROOT, 'lcgp_amd', 'csrc', '_hstamp_tmp.hip')
    open(tmp, 'w').write(src)
    try:
        subprocess.check_call(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-o', OUT, tmp])
    finally:
        os.remove(tmp)
    print('built', OUT)


def run(argv):
    os.environ['LCGP_HIP_LIB'] = OUT
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from lcgp_amd import LCGP, synth, _hip
    q = int(argv[0]) if argv else 8
    x, y, cfg = synth.make_config(3, q=q)
    m = LCGP(y=y, x=x, q=q)
    eng = m._get_engine()
    sc = _hip.default_sched()
    sc.hosted = 1
    for kv in argv[1:]:
        k, v = kv.split('=')
        setattr(sc, k, int(v))
    eng.sched = sc
    sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
    theta = m._theta_rows(sig_eff)
    lib = _hip.load()
    lib.lcgp_debug_host_stamps.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 256)()
    for _ in range(3):
        eng.evaluate(theta)
    torch.cuda.synchronize()
    lib.lcgp_debug_host_stamps(None, 1)
    eng.evaluate(theta)
    torch.cuda.synchronize()
    lib.lcgp_debug_host_stamps(buf, 0)
    st = np.array(list(buf), dtype=np.uint64).reshape(64, 4)
    print('q_local = %d, n = %d; s_memrealtime ticks of 10 ns' % (q, x.shape[0]))
    print('panel  chain workgroup (us)  last hosted tile ends (us after the chain started)  hosted tiles')
    for p in range(64):
        if st[p, 0] == 0:
            continue
        chain = (int(st[p, 1]) - int(st[p, 0])) / 100.0
        tiles = (int(st[p, 2]) - int(st[p, 0])) / 100.0 if st[p, 3] else 0.0
        print('%5d  %10.1f  %10.1f  %8d' % (p, chain, tiles, int(st[p, 3])))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    else:
        run(sys.argv[2:])
