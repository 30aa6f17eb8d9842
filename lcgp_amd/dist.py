"""Component-parallel sharding of the hot path.

The q latent components are independent (the reference fans them out over joblib threads,
lcgp.py:718-720 / 792-794).  Here component k lives on rank k mod G; every rank holds the small replicated
inputs; one all-reduce(sum) of a (P+1)-vector per evaluation is the only exchange (RCCL over xGMI when the
process group's backend is "nccl"; gloo in the CPU tests).  Every rank then runs the identical L-BFGS-B step
on identical numbers, so no broadcast of the next iterate is needed.
"""
from __future__ import annotations

import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def is_distributed(group=None) -> bool:
    try:
        dist = _dist()
    except Exception:
        return False
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def rank_world(group=None):
    if not is_distributed(group):
        return 0, 1
    dist = _dist()
    return dist.get_rank(group), dist.get_world_size(group)


def local_components(q: int, rank: int, world: int):
    """component k -> rank k mod world."""
    return list(range(rank, q, world))


def _all_reduce_impl(vec, group=None, device=None):
    """The collective itself (also exercised on a 1-rank RCCL group by tests/test_gpu_api.py)."""
    import torch
    dist = _dist()
    t = torch.as_tensor(np.ascontiguousarray(vec, np.float64))
    backend = dist.get_backend(group)
    if backend == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def all_reduce_sum(vec, group=None, device=None):
    """Sum a float64 numpy vector over the ranks (returns a new numpy array, identical on all ranks)."""
    if not is_distributed(group):
        return np.asarray(vec, np.float64)
    return _all_reduce_impl(vec, group, device)


def gather_rows(local_rows, q: int, group=None, device=None):
    """Assemble a (q, m) array from per-rank rows of the components `local_components(q, rank, world)`."""
    rank, world = rank_world(group)
    local_rows = np.asarray(local_rows, np.float64)
    m = local_rows.shape[1] if local_rows.ndim == 2 else 0
    full = np.zeros((q, m), np.float64)
    ks = local_components(q, rank, world)
    if len(ks):
        full[ks] = local_rows
    if world == 1:
        return full
    # disjoint rows: a sum is a gather
    return all_reduce_sum(full.reshape(-1), group, device).reshape(q, m)


def _broadcast_impl(arr, src=0, group=None, device=None):
    import torch
    dist = _dist()
    t = torch.as_tensor(np.ascontiguousarray(arr, np.float64)).clone()
    if dist.get_backend(group) == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return t.cpu().numpy()


def broadcast_array(arr, src=0, group=None, device=None):
    """Every rank returns rank `src`'s float64 array (used so that all ranks share ONE SVD basis)."""
    if not is_distributed(group):
        return np.asarray(arr, np.float64)
    return _broadcast_impl(arr, src, group, device)
