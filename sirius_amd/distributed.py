"""Multi-GPU MSM: one process per GPU, keys sharded block-cyclically (srs_ck_create_sharded); every rank
computes the partial commitment over its stripes, partials (64 B each) are all-gathered with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) and summed on
the host -- RCCL has no elliptic-curve reduction op, so the exchange is an all-gather of raw bytes
(SURVEY.md 8e).  Row-sharded structures (PlonkStructure.set_shard) return PARTIAL ProtoGalaxy polynomials: the same exchange
for a few dozen field elements (all_gather_field_sum)."""
import numpy as np

from .commitment import point_sum


def all_gather_commitments(curve, partial, group=None, device=None):
    """partial: (8,) or (m, 8) uint64 partial commitment(s) of this rank -> full commitment(s) on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return partial
    world = dist.get_world_size(group)      # a group of ONE rank still goes through the collective (and the host sum normalises)
    p = np.ascontiguousarray(partial, dtype=np.uint64).reshape(-1, 8)
    t = torch.from_numpy(p.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    g = torch.stack(outs).cpu().numpy().view(np.uint64)         # (world, m, 8)
    res = np.stack([point_sum(curve, g[:, j, :]) for j in range(p.shape[0])])
    return res if np.ndim(partial) == 2 else res[0]


def all_gather_field_sum(field, partial, group=None, device=None):
    """partial: (m, 4) uint64 Montgomery elements of `field` (this rank's partial polynomial / value) -> their sum over the
    ranks, on every rank.  There is no modular-sum reduction op either: all-gather of the raw bytes, sum through the library."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return partial
    from .protogalaxy import fold_witness
    from .field import ints_to_mont
    world = dist.get_world_size(group)
    p = np.ascontiguousarray(partial, dtype=np.uint64).reshape(-1, 4)
    t = torch.from_numpy(p.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    parts = [np.ascontiguousarray(o.cpu().numpy().view(np.uint64)) for o in outs]
    res = fold_witness(field, parts, ints_to_mont(field, [1] * world))       # srs_fold_lincomb on host vectors, coefficients 1
    return res.reshape(np.shape(partial))
