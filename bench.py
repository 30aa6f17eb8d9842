#!/usr/bin/env python
"""LCGP hot-path benchmark: NLL + gradient evaluations per second at BASELINE.json's headline configuration
(n=4096, d=6, p=64 -> q=8 latent components, fp64), on N GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 3
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
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=3, help='synthetic config id of lcgp_amd/synth.py (3 = headline)')
    ap.add_argument('--n', type=int, default=None, help='override n (debug only; invalidates the headline metric)')
    ap.add_argument('--q', type=int, default=None, help='override q (debug only: e.g. one rank\'s share of the components)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stages', action='store_true')
    ap.add_argument('--no-fit', action='store_true', help='skip the wall-clock fit() leg')
    ap.add_argument('--blocks', type=int, default=3, help='timed blocks of --steps evaluations (the first one is `value`)')
    ap.add_argument('--cpu-repeats', type=int, default=5)
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
    clk = torch.zeros(2, dtype=torch.int64, device=eng.device)
    _hip.check(lib.lcgp_lauum_clock(sp, *args, ws, C.c_void_p(clk.data_ptr())), 'lauum_clock')     # (read and clear: drops stale words)
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record(st)
        _hip.check(lib.lcgp_kernel_build(sp, eng.dtype, eng.kernel_id, *args[1:], C.c_void_p(eng.x.data_ptr()), C.c_void_p(0 if eng.sr is None else eng.sr.data_ptr()),
                                         C.c_void_p(eng.theta_dev.data_ptr()), ws), 'build')
        ev[1].record(st)
        _hip.check(lib.lcgp_potrf_logdet(sp, *args, ws, None, None, eng._sched(), eng.plan(False)), 'potrf')
        ev[2].record(st)
        _hip.check(lib.lcgp_trtri(sp, *args, ws, None), 'trtri')
        ev[3].record(st)
        _hip.check(lib.lcgp_lauum(sp, *args, ws, None), 'lauum')
        ev[4].record(st)
        torch.cuda.synchronize(eng.device)
        for i, k in enumerate(names):
            acc[k].append(ev[i].elapsed_time(ev[i + 1]))
    out = {k: float(np.median(v)) for k, v in acc.items()}
    # the clock the chip held during the last A^-1 launch above (un-profiled): its first workgroup stamps its K loop with the
    # shader-clock counter and the 100 MHz real-time counter (lcgp_lauum_clock)
    _hip.check(lib.lcgp_lauum_clock(sp, *args, ws, C.c_void_p(clk.data_ptr())), 'lauum_clock')
    cyc, ticks = (int(v) for v in clk.cpu())
    return out, dict(clock_mhz=100.0 * cyc / ticks if ticks > 0 else None, clock_window_us=ticks / 100.0)


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


def cpu_baseline(m, budget_s=12.0, repeats=5):
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
    repeats = max(1, int(repeats))
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
    if world > ndev and args.backend == 'nccl':
        # one rank per GPU or nothing: RCCL ranks folded onto one device would time something else than --gpus N
        raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible; the nccl (RCCL) backend needs one device per rank '
                         '(a single-GPU rehearsal of the launch line is `--backend gloo`)' % (world, ndev))
    dev_index = local_rank % ndev          # (more ranks than devices: gloo rehearsals on one GPU only, see above)
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

    def timed_block():
        """EXACTLY --steps evaluations between two barrier + synchronize brackets; (local seconds, max over ranks)."""
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            m.loss_and_grad(pts[i % len(pts)])
        barrier()
        dt_local = time.perf_counter() - t0
        dt_max = dt_local
        if world > 1:
            tt = torch.tensor([dt_local], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_max = float(tt.item())
        return dt_local, dt_max

    # block 0 is the contract's measurement (`value`, `ms_per_step`); the further blocks only show the spread
    blocks = [timed_block() for _ in range(max(1, args.blocks))]
    dt_local, dt = blocks[0]
    per_block_ms = [1e3 * b[1] / args.steps for b in blocks]
    log('timed %d block(s) of %d steps: %s ms/step' % (len(blocks), args.steps, ['%.3f' % v for v in per_block_ms]))
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
    out['ms_per_step_blocks'] = dict(values=per_block_ms, median=float(np.median(per_block_ms)), min=float(np.min(per_block_ms)),
                                     max=float(np.max(per_block_ms)),
                                     note='block 0 is `value`/`ms_per_step`; every block = --steps evaluations, max over ranks')
    out['path_tflops'] = flops_eval / (dt / args.steps) / 1e12
    out['path_frac_of_mfma_peak'] = out['path_tflops'] / (peak * world)
    out['library'] = dict(source_hash=_hip.loaded_hash(), version=int(_hip.load().lcgp_version()))

    # ---- per-rank stage timings (every rank that holds components times its own kernels with HIP events) ----
    ql = len(m._local_ks)
    mine = dict(rank=rank, local_rank=local_rank, device_index=dev_index, q_local=ql, components=list(m._local_ks),
                ms_per_step=1e3 * dt_local / args.steps, host=os.uname().nodename, pid=os.getpid())
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        mine['device_name'] = pr.name
        mine['pci_bus_id'] = '%04x:%02x:%02x' % (getattr(pr, 'pci_domain_id', 0), getattr(pr, 'pci_bus_id', -1) & 0xff,
                                                  getattr(pr, 'pci_device_id', 0))
        mine['device_uuid'] = str(getattr(pr, 'uuid', ''))
    except Exception as e:      # never lose the line over a missing attribute
        mine['device_name'] = 'unknown (%s)' % e
    if not args.no_stages:
        m.loss_and_grad(pts[0])          # collective: every rank takes part (leaves theta_0 resident)
    st = clock = None
    if not args.no_stages and m._engine is not None:
        st, clock = stage_times(m)
        mine['lauum_clock'] = clock
        esz = 8 if dtype == 'float64' else 4
        mine['stages_ms'] = st
        mine['stage_tflops'] = {k: ql * n ** 3 / 3.0 / (st[k] * 1e-3) / 1e12 for k in ('potrf', 'trtri', 'lauum')}
        mine['stage_frac_of_peak'] = {k: v / peak for k, v in mine['stage_tflops'].items()}
        mine['build_gbs'] = ql * (n * n / 2.0) * esz / (st['build'] * 1e-3) / 1e9
        mine['path_tflops'] = ql * float(n) ** 3 / (dt_local / args.steps) / 1e12
        mine['path_frac_of_mfma_peak'] = mine['path_tflops'] / peak

    # ---- the collective on its own: the all-reduce of the (P+3)-vector, device-resident, timed alone ----
    allreduce_us = None
    if world > 1:
        width = 3 + q * d + 2 * q + p
        t = torch.zeros(width, dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        for _ in range(20):
            dist.all_reduce(t)
        barrier()
        samples = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(100):
                dist.all_reduce(t)
            torch.cuda.synchronize()
            samples.append((time.perf_counter() - t0) / 100 * 1e6)
        allreduce_us = float(np.median(samples))
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    out['dist'] = dict(backend=(str(dist.get_backend()) if world > 1 else 'none (single rank: no collective on the path)'),
                       world_size=(dist.get_world_size() if world > 1 else 1), launched_gpus=args.gpus,
                       visible_devices=ndev, allreduce_us=allreduce_us,
                       allreduce_note='median of 5 x 100 in-place all_reduce(sum) of the %d-double vector, alone on the stream'
                                      % (3 + q * d + 2 * q + p) if world > 1 else None,
                       distinct_devices=len({(r.get('host'), r.get('pci_bus_id'), r.get('device_index')) for r in ranks}),
                       ranks=ranks)

    # ---- wall-clock fit() of the same configuration (the metric names it; lcgp.py:537-540), all ranks in lock-step ----
    if not args.no_fit:
        mf = LCGP(y=y, x=x, q=cfg['q'], submethod=cfg['submethod'], dtype=dtype, device='cuda:%d' % dev_index)
        mf._get_engine()
        barrier()
        t0 = time.perf_counter()
        mf.fit()
        barrier()
        t_fit = time.perf_counter() - t0
        x0 = np.random.default_rng(11).uniform(0.0, 1.0, (2000, d))
        t0 = time.perf_counter()
        mf.predict(x0)
        barrier()
        t_pred = time.perf_counter() - t0
        res = mf.opt_result
        out['fit'] = dict(fit_wallclock_s=t_fit, iterations=int(res.nit), evaluations=int(res.nfev), final_loss=float(res.fun),
                          converged=bool(res.success), message=str(res.message),
                          predict_2000_wallclock_s=t_pred,
                          restarts=len(getattr(res, 'restarts', [])) - 1,
                          float32_fallbacks=(int(res.float32_fallbacks) if dtype == 'float32' else None),
                          float64_only=(bool(res.float64_only) if dtype == 'float32' else None),
                          ms_per_evaluation=1e3 * t_fit / max(int(res.nfev), 1),
                          note='LCGP(...).fit() from the initial parameters (scipy L-BFGS-B, defaults) on this run\'s ranks; '
                               'the reference algorithm needs one cpu_baseline evaluation per L-BFGS-B evaluation')
        log('fit: %.3f s, %d iterations, %d evaluations, final loss %.6g' % (t_fit, res.nit, res.nfev, res.fun))
        del mf

    if rank == 0 and st is not None:
        clock_mhz, clock_window_us = clock['clock_mhz'], clock['clock_window_us']
        log('stages (ms): %s; shader clock during A^-1 = W^T W: %s MHz over %s us' % (st, clock_mhz and round(clock_mhz), clock_window_us))
        out['stages_ms'] = st
        # dominant kernel: the single-launch LAUUM (tile_gemm<OP_LAUUM>): A^-1 = W^T W, n^3/3 flops per component
        fl = ql * float(n) ** 3 / 3.0
        # HBM-side bytes of that launch come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X guide),
        # summarised in profiles/traffic.json together with the source hash of the library they were measured on:
        # a number measured on another build of the kernels is NOT reported
        traffic = mfma_busy = None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
            if world == 1 and args.config == 3 and args.n is None and args.q is None and tj.get('lib_hash') == _hip.loaded_hash():
                traffic = tj.get('tile_gemm_lauum_bytes_per_launch')
                mfma_busy = tj.get('tile_gemm_lauum_mfma_busy')
        except Exception:
            pass
        sched = _hip.default_sched()
        nb2 = (n + 127) // 128
        progressive = ql * nb2 * (nb2 + 1) // 2 <= sched.progressive_tiles
        out['roofline'] = dict(bound='mfma', kernel='tile_gemm<double, OP_LAUUM> (A^-1 = W^T W with z = A^-1 b in its epilogue, one launch per evaluation; '
                                                    'lcgp_lauum enqueues the same launch)',
                               achieved=fl / (st['lauum'] * 1e-3) / 1e12, peak=peak, unit='TFLOP/s',
                               frac=fl / (st['lauum'] * 1e-3) / 1e12 / peak, traffic=traffic,
                               flops_per_launch=fl, launch_ms=st['lauum'],
                               clock_mhz=clock_mhz, clock_window_us=clock_window_us,
                               clock_note='shader-clock cycles / real time of the longest K loop of THIS run\'s last A^-1 launch '
                                          '(in-kernel s_memtime / s_memrealtime, un-profiled); the 78.6 TFLOP/s peak is 256 CUs x 128 flop/cycle x 2400 MHz',
                               mfma_busy=mfma_busy,
                               mfma_busy_note='SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of this kernel from the '
                                              'rocprofv3 --pmc pass in profiles/ (same library hash), or null',
                               launched_by_the_timed_path=not progressive,
                               note=None if not progressive else
                               'with %d component(s) on this rank the timed path forms the inverse behind the factorisation '
                               '(progressive filler jobs), not with this launch; it is timed here on its own' % ql)
        # the stage groups beside it, each against the same peak (the factorisation is the one furthest below it)
        out['roofline_stages'] = {k: dict(bound='mfma', flops=ql * n ** 3 / 3.0, ms=st[k], achieved=mine['stage_tflops'][k],
                                          peak=peak, unit='TFLOP/s', frac=mine['stage_frac_of_peak'][k])
                                  for k in ('potrf', 'trtri', 'lauum')}
        out['stage_tflops'] = mine['stage_tflops']
        out['build_gbs'] = mine['build_gbs']     # lower tiles only are written
        if args.predict > 0:
            # (the stage passes above re-ran build / factorisation / inverse at the resident theta_0: L^-1 and z are consistent)
            pr = predict_leg(m, args.predict)
            pr['frac_of_mfma_peak'] = pr['tflops'] / peak
            out['predict'] = pr
            log('predict leg: n0=%d %.3f ms = %.1f TFLOP/s' % (pr['n0'], pr['ms'], pr['tflops']))

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('cpu baseline (bounded sample) ...')
        base, pieces, ns = cpu_baseline(m, repeats=args.cpu_repeats)
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
    if rank == 0 and args.q is not None and world == 1:
        # `--q Q` runs ONE rank's share of the configuration on this one GPU: with component k on rank k mod G that is the
        # load of every rank of a G = q_config / Q GPU job, so the line also says what it projects to (an all-reduce of
        # ~1 KB per evaluation is the only thing a real job adds).  A PROJECTION, printed so that the first measured
        # SCALE record can be read against it -- never a measured multi-GPU number.
        q_full = int(synth.CONFIGS[args.config]['q'])
        # every GPU count whose SLOWEST rank holds this many components (k -> rank k mod G: ceil(q / G) of them): with q = 6 on
        # 4 GPUs the shares are 2 / 2 / 1 / 1 and the step is the 2-component share's
        gs = [g for g in range(1, q_full + 1) if -(-q_full // g) == args.q]
        if gs:
            out['projected_from_q_local'] = dict(q_local=args.q, n_gpus=gs[0], n_gpus_with_this_slowest_share=gs,
                                                 ms_per_step=out['ms_per_step'], evals_per_s=1e3 / out['ms_per_step'],
                                                 note='projection from a one-GPU run of the slowest rank\'s share; not a measured '
                                                      'multi-GPU number')
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
