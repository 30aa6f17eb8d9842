"""Component-parallel sharding of the hot path.

The q latent components are independent (the reference fans them out over joblib threads,
lcgp.py:718-720 / 792-794).  Here component k lives on rank k mod G; every rank holds the small replicated
inputs; one all-reduce(sum) of a (P+2)-vector per evaluation is the only exchange.  With the `nccl` backend
(= RCCL over xGMI) the vector is assembled on the device by the library (lcgp_pack_partial), reduced in
place and copied to the host ONCE; with `gloo` (the CPU tests) the same calls run on host tensors.  Every rank
then runs the identical L-BFGS-B step on identical numbers, so no broadcast of the next iterate is needed.

Collectives are used when the job has more than one rank, or when a process group was passed explicitly
(`LCGP(..., process_group=g)`): a one-rank RCCL group then still goes through every collective, which is how
the production path is exercised on a single-GPU test box.
"""
from __future__ import annotations

import threading

import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def _initialized() -> bool:
    try:
        dist = _dist()
    except Exception:
        return False
    return dist.is_available() and dist.is_initialized()


def is_distributed(group=None) -> bool:
    return _initialized() and _dist().get_world_size(group) > 1


def use_collectives(group=None) -> bool:
    """More than one rank, or an explicitly passed process group (even of one rank)."""
    return _initialized() and (group is not None or _dist().get_world_size() > 1)


def backend_is_nccl(group=None) -> bool:
    return _initialized() and str(_dist().get_backend(group)) == "nccl"


def rank_world(group=None):
    if not _initialized() or not use_collectives(group):
        return 0, 1
    dist = _dist()
    return dist.get_rank(group), dist.get_world_size(group)


def local_components(q: int, rank: int, world: int):
    """component k -> rank k mod world."""
    return list(range(rank, q, world))



# per host thread: device index -> ONE growable pinned float64 row for the device-to-host copy of an evaluation / prediction
# (thread-local: the stream synchronisation below releases the GIL, and two models evaluated from two threads on one device
# would otherwise overwrite each other's row between the copy and the read)
_TLS = threading.local()


def reduce_to_host(t, group=None):
    """Sum the float64 torch tensor `t` over the ranks IN PLACE where it lives (device tensor + nccl: RCCL, no host
    round trip before the collective) and return it as a numpy array: exactly one device-to-host copy."""
    import torch
    if use_collectives(group):
        dist = _dist()
        if backend_is_nccl(group):
            if not t.is_cuda:
                t = t.to(torch.device("cuda", torch.cuda.current_device()))
        elif t.is_cuda:
            t = t.cpu()          # gloo rehearsal of a GPU job: the collective runs on host memory
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    if t.is_cuda:
        # one asynchronous copy into the device's pinned row (reallocated only when a longer vector comes along: the page-locked
        # memory of a serving loop with varying prediction batch sizes stays bounded by the largest one) and one stream
        # synchronisation: no pageable staging buffer and no allocation per evaluation
        n = t.numel()
        rows = getattr(_TLS, 'rows', None)
        if rows is None:
            rows = _TLS.rows = {}
        pin = rows.get(t.device.index)
        if pin is None or pin.numel() < n:
            pin = rows[t.device.index] = torch.empty(max(n, 256), dtype=torch.float64).pin_memory()
        pin[:n].copy_(t.reshape(-1), non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return pin[:n].numpy().reshape(tuple(t.shape)).copy()
    return t.cpu().numpy()


def all_reduce_sum(vec, group=None, device=None):
    """Sum a float64 numpy vector over the ranks (returns a new numpy array, identical on all ranks)."""
    import torch
    t = torch.as_tensor(np.ascontiguousarray(vec, np.float64)).clone()
    if use_collectives(group) and backend_is_nccl(group):
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    return reduce_to_host(t, group)


def gather_rows(local_rows, q: int, group=None, device=None):
    """Assemble a (q, m) array from per-rank rows of the components `local_components(q, rank, world)`: ONE all_gather of
    the local rows (every rank sends ceil(q / world) rows, padded; nothing is summed and no rank ships zeros for the
    components it does not hold -- round 3 summed q x m zero-padded blocks with an all-reduce, 1 GB for the (q, n, n)
    cache views at the headline size)."""
    rank, world = rank_world(group)
    local_rows = np.asarray(local_rows, np.float64)
    m = local_rows.shape[1] if local_rows.ndim == 2 else 0
    full = np.zeros((q, m), np.float64)
    ks = local_components(q, rank, world)
    if not use_collectives(group):
        if len(ks):
            full[ks] = local_rows
        return full
    import torch
    dist = _dist()
    per = (q + world - 1) // world                     # rows a rank holds at most (component k -> rank k mod world)
    mine = np.zeros((per, m), np.float64)
    if len(ks):
        mine[:len(ks)] = local_rows
    t = torch.as_tensor(mine)
    if backend_is_nccl(group):
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    out = torch.empty((world * per, m), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    got = out.cpu().numpy().reshape(world, per, m)
    for r in range(world):
        rk = local_components(q, r, world)
        if len(rk):
            full[rk] = got[r, :len(rk)]
    return full


def broadcast_array(arr, src=0, group=None, device=None):
    """Every rank returns rank `src`'s float64 array (used so that all ranks share ONE SVD basis)."""
    if not use_collectives(group):
        return np.asarray(arr, np.float64)
    import torch
    dist = _dist()
    t = torch.as_tensor(np.ascontiguousarray(arr, np.float64)).clone()
    if backend_is_nccl(group):
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return t.cpu().numpy()
