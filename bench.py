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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=3, help='synthetic config id of lcgp_amd/synth.py (3 = headline)')
    ap.add_argument('--n', type=int, default=None, help='override n (debug only; invalidates the headline metric)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stages', action='store_true')
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
        _hip.check(lib.lcgp_potrf_logdet(sp, *args, ws, None, None), 'potrf')
        ev[2].record(st)
        _hip.check(lib.lcgp_trtri(sp, *args, ws), 'trtri')
        ev[3].record(st)
        _hip.check(lib.lcgp_lauum(sp, *args, ws), 'lauum')
        ev[4].record(st)
        torch.cuda.synchronize(eng.device)
        for i, k in enumerate(names):
            acc[k].append(ev[i].elapsed_time(ev[i + 1]))
    return {k: float(np.median(v)) for k, v in acc.items()}


def cpu_baseline(m, x, y, cfg):
    """Reference algorithm (eigh form + autodiff) and Cholesky form, ONE component of the same workload."""
    from oracle import cpu_baseline as cb
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    lLmb, lLmb0, ls2b, lnug = (t.numpy() for t in m.get_param())
    phi, D = m.phi.numpy(), m.diag_D.numpy()
    xs, ys = m.x.numpy(), m.y.numpy()
    k = 0
    t_eigh, v_eigh, g_eigh = cb.eigh_form_component(xs, ys, phi[:, k], D[k], lLmb[k], lLmb0[k], lnug[k], ls2b, threads=cores)
    t_chol, pieces = cb.chol_form_component(xs, ys, phi[:, k], D[k], lLmb[k], lLmb0[k], lnug[k], ls2b)
    q = int(m.q)
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except Exception:
        pass
    return dict(value=1.0 / (q * t_eigh), unit='evals/s', cores=cores, kind='port',
                sample='1 of %d components at n=%d (eigh-form NLL + autodiff gradient, torch CPU fp64, %.1f s), '
                       'scaled by q' % (q, int(m.n), t_eigh),
                cpu_model=model, eigh_form_s_per_component=t_eigh, chol_form_s_per_component=t_chol,
                chol_form_evals_per_s=1.0 / (q * t_chol)), pieces, (v_eigh, g_eigh)


def main():
    args = parse()
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the hot path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    from lcgp_amd import LCGP, synth
    over = {} if args.n is None else dict(n=args.n)
    x, y, cfg = synth.make_config(args.config, **over)
    dtype = 'float64' if cfg['dtype'] == 'f64' else 'float32'
    m = LCGP(y=y, x=x, q=cfg['q'], submethod=cfg['submethod'], dtype=dtype, device='cuda:%d' % local_rank)
    pts = synth.param_points(args.config, m._get_flat())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
        tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

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

    if rank == 0 and not args.no_stages and m._engine is not None:
        m.loss_and_grad(pts[0])
        st = stage_times(m)
        ql = len(m._local_ks)
        out['stages_ms'] = st
        # dominant kernel: the single-launch LAUUM (tile_gemm<OP_LAUUM>): A^-1 = W^T W, n^3/3 flops per component
        fl = ql * float(n) ** 3 / 3.0
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
            traffic = tj.get('tile_gemm_lauum_bytes_per_launch')
        except Exception:
            pass
        out['roofline'] = dict(bound='mfma', kernel='tile_gemm<double, OP_LAUUM> (A^-1 = W^T W, one launch per evaluation)',
                               achieved=fl / (st['lauum'] * 1e-3) / 1e12, peak=peak, unit='TFLOP/s',
                               frac=fl / (st['lauum'] * 1e-3) / 1e12 / peak, traffic=traffic,
                               flops_per_launch=fl, launch_ms=st['lauum'])
        out['stage_tflops'] = dict(potrf=ql * n ** 3 / 3.0 / (st['potrf'] * 1e-3) / 1e12,
                                   trtri=ql * n ** 3 / 3.0 / (st['trtri'] * 1e-3) / 1e12,
                                   lauum=fl / (st['lauum'] * 1e-3) / 1e12)
        esz = 8 if dtype == 'float64' else 4
        out['build_gbs'] = ql * (n * n / 2.0) * esz / (st['build'] * 1e-3) / 1e9     # lower tiles only are written

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        m.loss_and_grad(pts[0])
        row = m._engine.out_dev.cpu().numpy()[0]
        base, pieces, _ = cpu_baseline(m, x, y, cfg)
        out['cpu_baseline'] = base
        # parity gate in the same run: component 0, GPU vs the Cholesky-form oracle (NLL 1e-6, gradient 1e-5)
        v_gpu = row[0] - row[1] / (2.0 * m.diag_D.numpy()[0])
        g_gpu = row[3:5 + d]
        g_ref = np.concatenate([pieces['g_ell'], [pieces['g_scale'], pieces['g_nug']]])
        out['parity'] = dict(nll_rel_err=abs(v_gpu - pieces['value']) / abs(pieces['value']),
                             grad_rel_err=float(np.max(np.abs(g_gpu - g_ref)) / np.max(np.abs(g_ref))),
                             passed=bool(abs(v_gpu - pieces['value']) <= 1e-6 * abs(pieces['value']) and
                                         np.max(np.abs(g_gpu - g_ref)) <= 1e-5 * np.max(np.abs(g_ref))))
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
