#!/usr/bin/env python
"""LCGP hot-path benchmark: NLL + gradient evaluations per second at BASELINE.json's headline configuration
(n=4096, d=6, p=64 -> q=8 latent components, fp64), on N GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete evaluation: host packs the constrained parameters, one H2D copy, the HIP path for
this rank's components (k -> rank k mod N), one all-reduce of the (P+1)-vector (RCCL) when N > 1, D2H, chain
rule -- exactly what `LCGP.fit()` hands L-BFGS-B per function evaluation.  Inputs are resident in HBM before
the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X fp64 matrix (= vector) peak, SURVEY.md 8(d) / BASELINE.md 4
FP32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
T_START = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=3, help='synthetic config id of lcgp_amd/synth.py (3 = headline)')
    ap.add_argument('--n', type=int, default=None, help='override n (debug only; invalidates the headline metric)')
    ap.add_argument('--q', type=int, default=None, help='override q (debug only: e.g. one rank\'s share of the components)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stages', action='store_true')
    ap.add_argument('--predict', type=int, default=2000, help='new inputs of the predict leg (K6); 0 = skip')
    ap.add_argument('--backend', default='nccl', help='process-group backend for --gpus > 1 (nccl = RCCL)')
    return ap.parse_args()


def stage_times(m, reps=3):
    """HIP-event timing (on the stream the kernels are launched on) of the stages of one evaluation."""
    import ctypes as C
    import torch
    from lcgp_amd import _hip
    eng = m._engine
    lib = eng.lib
    st = torch.cuda.current_stream(eng.device)
    sp = C.c_void_p(st.cuda_stream)
    args = (eng.dtype, eng.n, eng.d, eng.p, eng.q_local)
    ws = C.c_void_p(eng.workspace.data_ptr())
    names = ['build', 'potrf', 'trtri', 'lauum']
    acc = {k: [] for k in names}
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record(st)
        _hip.check(lib.lcgp_kernel_build(sp, *args, C.c_void_p(eng.x.data_ptr()), C.c_void_p(0 if eng.sr is None else eng.sr.data_ptr()),
                                         C.c_void_p(eng.theta_dev.data_ptr()), ws), 'build')
        ev[1].record(st)
        _hip.check(lib.lcgp_potrf_logdet(sp, *args, ws, None, None, None), 'potrf')
        ev[2].record(st)
        _hip.check(lib.lcgp_trtri(sp, *args, ws, None), 'trtri')
        ev[3].record(st)
        _hip.check(lib.lcgp_lauum(sp, *args, ws, None), 'lauum')
        ev[4].record(st)
        torch.cuda.synchronize(eng.device)
        for i, k in enumerate(names):
            acc[k].append(ev[i].elapsed_time(ev[i + 1]))
    return {k: float(np.median(v)) for k, v in acc.items()}


def predict_leg(m, n0, reps=5):
    """K6 (lcgp.py:808-859): latent mean / variance of n0 new inputs from the factorisation in the workspace, timed with
    HIP events on the launch stream.  Work per call: U_k = X_k W_k^T, W lower triangular: n0 n^2 flops per component."""
    import torch
    eng = m._engine
    rng = np.random.default_rng(7)
    x0s = rng.uniform(0.0, 1.0, (n0, int(m.d)))
    st = torch.cuda.current_stream(eng.device)
    eng.predict_device(x0s)                                  # warm-up (allocates the engine-owned scratch)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        eng.predict_device(x0s)
        e1.record(st)
        torch.cuda.synchronize(eng.device)
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    fl = float(eng.q_local) * n0 * float(eng.n) ** 2
    return dict(n0=n0, ms=ms, flops=fl, tflops=fl / (ms * 1e-3) / 1e12,
                note='cross covariance + U = X W^T on the MFMA tile kernel + row reductions, all local components per launch; '
                     'includes the H2D copy of x0')


def log(msg):
    print('[bench %.1fs] %s' % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def host_cores():
    """CPU share of this process: affinity mask capped by the cgroup quota (the GPU box gives 16 of many)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except Exception:
        pass
    if os.environ.get('LCGP_CPU_THREADS'):
        cores = int(os.environ['LCGP_CPU_THREADS'])
    return cores


def cpu_baseline(m, budget_s=12.0, repeats=3):
    """Reference algorithm (eigh form + autodiff) and Cholesky form for ONE component of the same workload on the
    host cores: median of `repeats` samples each, after one warm-up probe.  Bounded: the probe at n=1024 picks the
    largest prefix of the training set (n, n/2, n/4 ...) whose predicted eigh-form time fits `budget_s` per sample; a
    shorter prefix is scaled by (n/n_s)^3 and reported as such."""
    from oracle import cpu_baseline as cb
    from threadpoolctl import threadpool_limits
    cores = host_cores()
    lLmb, lLmb0, ls2b, lnug = (t.numpy() for t in m.get_param())
    phi, D = m.phi.numpy(), m.diag_D.numpy()
    xs, ys = m.x.numpy(), m.y.numpy()
    n, q, k = int(m.n), int(m.q), 0

    def eigh(ns):
        with threadpool_limits(limits=cores):
            return cb.eigh_form_component(xs[:ns], ys[:, :ns], phi[:, k], D[k], lLmb[k], lLmb0[k], lnug[k], ls2b, threads=cores)

    def chol(ns):
        with threadpool_limits(limits=cores):
            return cb.chol_form_component(xs[:ns], ys[:, :ns], phi[:, k], D[k], lLmb[k], lLmb0[k], lnug[k], ls2b, threads=cores)
    probe_n = min(n, 1024)
    t_probe = eigh(probe_n)[0]                     # also the warm-up (thread pools, MKL initialisation)
    chol(probe_n)
    log('cpu probe: eigh-form component at n=%d took %.2f s on %d threads' % (probe_n, t_probe, cores))
    ns = n
    while ns > probe_n and t_probe * (ns / probe_n) ** 3 > budget_s:
        ns //= 2
    t_eighs = [eigh(ns)[0] for _ in range(repeats)]
    chols = [chol(ns) for _ in range(repeats)]
    t_chols = [c[0] for c in chols]
    pieces = chols[-1][1]
    t_eigh, t_chol = float(np.median(t_eighs)), float(np.median(t_chols))
    log('cpu samples: n_s=%d eigh-form %s s, chol-form %s s' % (ns, ['%.2f' % t for t in t_eighs], ['%.2f' % t for t in t_chols]))
    scale = (n / ns) ** 3
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except Exception:
        pass
    sample = ('median of %d timings of 1 of %d components, first %d of %d inputs (eigh-form NLL + autodiff gradient, torch CPU '
              'fp64, %d threads, %.1f s each)' % (repeats, q, ns, n, cores, t_eigh))
    sample += ', scaled by q' + ('' if ns == n else ' and by (n/n_s)^3 = %g' % scale)
    base = dict(value=1.0 / (q * t_eigh * scale), unit='evals/s', cores=cores, kind='port', sample=sample,
                cpu_model=model, n_sample=ns, repeats=repeats,
                eigh_form_s_per_component=t_eigh * scale, eigh_form_samples_s=[t * scale for t in t_eighs],
                chol_form_s_per_component=t_chol * scale, chol_form_samples_s=[t * scale for t in t_chols],
                chol_form_evals_per_s=1.0 / (q * t_chol * scale))
    return base, pieces, ns


def _lk(m):
    m._get_engine()
    return m._local_ks


def main():
    args = parse()
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the hot path has no CPU fallback)')
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # (more ranks than devices only happens in single-GPU rehearsals)
    torch.cuda.set_device(dev_index)
    import torch.distributed as dist
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    from lcgp_amd import LCGP, synth, _hip
    over = {} if args.n is None else dict(n=args.n)
    if args.q is not None:
        over['q'] = args.q
    x, y, cfg = synth.make_config(args.config, **over)
    dtype = 'float64' if cfg['dtype'] == 'f64' else 'float32'
    m = LCGP(y=y, x=x, q=cfg['q'], submethod=cfg['submethod'], dtype=dtype, device='cuda:%d' % dev_index)
    pts = synth.param_points(args.config, m._get_flat())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log('model built: n=%d d=%d p=%d q=%d, %d local components' % (int(m.n), int(m.d), int(m.p), int(m.q), len(_lk(m))))
    last = None
    for i in range(args.warmup):
        last = m.loss_and_grad(pts[i % len(pts)])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = m.loss_and_grad(pts[i % len(pts)])
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    log('timed %d steps: %.3f ms/step' % (args.steps, 1e3 * dt / args.steps))
    n, d, p, q = int(m.n), int(m.d), int(m.p), int(m.q)
    flops_eval = float(q) * float(n) ** 3                       # potrf n^3/3 + inverse 2n^3/3 per component
    peak = FP64_MFMA_PEAK_TFLOPS if dtype == 'float64' else FP32_MFMA_PEAK_TFLOPS
    out = dict(metric='NLL+grad evals/sec', value=args.steps / dt, unit='evals/s', n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling='strong',
               vs_baseline=None, dtype='f64' if dtype == 'float64' else 'f32', data='synthetic',
               config=dict(workload='configs[%d]: n=%d d=%d p=%d q=%d submethod=%s, one NLL+gradient evaluation per step'
                           % (args.config - 1, n, d, p, q, cfg['submethod']),
                           n=n, d=d, p=p, q=q, parallelism='latent components k -> rank k mod %d' % world,
                           q_local_rank0=len(m._local_ks)))
    out['path_tflops'] = flops_eval / (dt / args.steps) / 1e12
    out['path_frac_of_mfma_peak'] = out['path_tflops'] / (peak * world)

    if not args.no_stages:
        m.loss_and_grad(pts[0])          # collective: every rank takes part (leaves theta_0 resident)
    if rank == 0 and not args.no_stages and m._engine is not None:
        st = stage_times(m)
        log('stages (ms): %s' % st)
        ql = len(m._local_ks)
        out['stages_ms'] = st
        # dominant kernel: the single-launch LAUUM (tile_gemm<OP_LAUUM>): A^-1 = W^T W, n^3/3 flops per component
        fl = ql * float(n) ** 3 / 3.0
        # HBM-side bytes of that launch come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X guide),
        # summarised in profiles/traffic.json together with the source hash of the library they were measured on:
        # a number measured on another build of the kernels is NOT reported
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
            if world == 1 and args.config == 3 and args.n is None and args.q is None and tj.get('lib_hash') == _hip.loaded_hash():
                traffic = tj.get('tile_gemm_lauum_bytes_per_launch')
        except Exception:
            pass
        out['roofline'] = dict(bound='mfma', kernel='tile_gemm<double, OP_LAUUM> (A^-1 = W^T W with z = A^-1 b in its epilogue, one launch per evaluation; '
                                                    'lcgp_lauum enqueues the same launch)',
                               achieved=fl / (st['lauum'] * 1e-3) / 1e12, peak=peak, unit='TFLOP/s',
                               frac=fl / (st['lauum'] * 1e-3) / 1e12 / peak, traffic=traffic,
                               flops_per_launch=fl, launch_ms=st['lauum'])
        out['stage_tflops'] = dict(potrf=ql * n ** 3 / 3.0 / (st['potrf'] * 1e-3) / 1e12,
                                   trtri=ql * n ** 3 / 3.0 / (st['trtri'] * 1e-3) / 1e12,
                                   lauum=fl / (st['lauum'] * 1e-3) / 1e12)
        esz = 8 if dtype == 'float64' else 4
        out['build_gbs'] = ql * (n * n / 2.0) * esz / (st['build'] * 1e-3) / 1e9     # lower tiles only are written
        if args.predict > 0:
            # (the stage passes above re-ran build / factorisation / inverse at the resident theta_0: L^-1 and z are consistent)
            pr = predict_leg(m, args.predict)
            pr['frac_of_mfma_peak'] = pr['tflops'] / peak
            out['predict'] = pr
            log('predict leg: n0=%d %.3f ms = %.1f TFLOP/s' % (pr['n0'], pr['ms'], pr['tflops']))

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('cpu baseline (bounded sample) ...')
        base, pieces, ns = cpu_baseline(m)
        out['cpu_baseline'] = base
        # parity gate in the same run, same sample: component 0, HIP path vs the Cholesky-form oracle
        # (NLL 1e-6 relative, gradient 1e-5 relative to max |g|)
        from lcgp_amd.engine import HotPathEngine
        eng = HotPathEngine(m.x.numpy()[:ns], m.y.numpy()[:, :ns], None, 1, dtype, 'cuda:%d' % dev_index)
        lLmb, lLmb0, ls2b, lnug = (t.numpy() for t in m.get_param())
        th = np.concatenate([lLmb[0], [lLmb0[0], lnug[0], m.diag_D.numpy()[0]], m.phi.numpy()[:, 0] / np.exp(0.5 * ls2b)])
        row = eng.evaluate(th[None, :])[0]
        v_gpu = row[0] - row[1] / (2.0 * m.diag_D.numpy()[0])
        g_gpu = row[3:5 + d]
        g_ref = np.concatenate([pieces['g_ell'], [pieces['g_scale'], pieces['g_nug']]])
        e_v = abs(v_gpu - pieces['value']) / abs(pieces['value'])
        e_g = float(np.max(np.abs(g_gpu - g_ref)) / np.max(np.abs(g_ref)))
        # fp64: BASELINE.json's tolerances; fp32 has no reference: this build's own stated tolerance
        tol_v, tol_g = (1e-6, 1e-5) if dtype == 'float64' else (1e-3, 1e-3)
        out['parity'] = dict(nll_rel_err=e_v, grad_rel_err=e_g, n_sample=ns, nll_tol=tol_v, grad_tol=tol_g,
                             passed=bool(e_v <= tol_v and e_g <= tol_g))
        del eng
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
