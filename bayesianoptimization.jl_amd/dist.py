"""Multi-GPU sharding of the multi-restart candidate set.

Candidates are independent, so rank g scores columns [g*R/G, (g+1)*R/G) with no data-path
collective.  The ONE exchange step is the arg-max: RCCL has no MAXLOC, so every rank contributes its
16-byte record (float64 value bits, int64 GLOBAL column index) to a single all_gather
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests) and every rank applies the same
(value desc, index asc) reduction -> bit-identical winner everywhere, identical to the 1-GPU result
(reference tie rule: strict '>' keeps the first maximum, src/acquisition.jl:62).

libbohip picks one of three summation schedules by batch size (row-wise, split-K, whole-K); a rank that scores a shard
calls ``model.set_batch_hint(R_total)`` first so that every shard is summed exactly like the unsharded batch.
"""
from __future__ import annotations

import math

import numpy as np


def shard_bounds(R, world, rank):
    """Contiguous column range of `rank` (last ranks get the remainder-free floor split + spill)."""
    base, rem = divmod(R, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_best(vals, idxs):
    """(value desc, index asc); NaN / -Inf / idx<0 never win.  Returns (-inf, -1) if nothing qualifies."""
    best_v, best_i = -math.inf, -1
    for v, i in zip(vals, idxs):
        i = int(i)
        if i < 0 or not (v > -math.inf):
            continue
        if best_i < 0 or v > best_v or (v == best_v and i < best_i):
            best_v, best_i = float(v), i
    return best_v, best_i


def allgather_best(rec, offset, world, force_collective=False):
    """rec: int64 tensor on this rank's device holding (value bits, local index[, global column offset]).
    With the 3-element form the offset travels inside the record, so the exchange is exactly ONE collective and
    no other device work; with the 2-element form `offset` is added on the device first.
    Returns the global (value, index) on every rank."""
    import torch

    has_off = rec.numel() >= 3
    if world == 1 and not force_collective:
        h = rec.cpu().numpy()
        v = float(h[:1].view(np.float64)[0])
        off = int(h[2]) if has_off else int(offset)
        return (v, int(h[1]) + off) if h[1] >= 0 else (-math.inf, -1)
    import torch.distributed as dist

    g = rec
    if not has_off:
        g = rec.clone()
        g[1] = torch.where(g[1] >= 0, g[1] + offset, g[1])
    if dist.get_backend() == "gloo":
        g = g.cpu()          # CPU tests and the shared-GPU test mode of bench.py
    n = g.numel()
    out = torch.empty(world * n, dtype=torch.int64, device=g.device)
    dist.all_gather_into_tensor(out, g)
    h = out.cpu().numpy().reshape(world, n)
    vals = h[:, 0].copy().view(np.float64)
    idxs = h[:, 1] + h[:, 2] * (h[:, 1] >= 0) if has_off else h[:, 1]
    return reduce_best(vals, idxs)


def allgather_best_many(vals, idxs, offset, world):
    """Thompson form (BASELINE configs[4]): this rank holds S per-draw records (vals float64[S], idxs int64[S], indices
    local to its candidate shard starting at global column `offset`).  ONE all_gather of S x 16 bytes per rank, then
    the same per-draw (value desc, index asc) reduction on every rank.  Tensors may live on any device."""
    import torch

    S = vals.numel()
    gidx = torch.where(idxs >= 0, idxs + offset, idxs)
    rec = torch.stack([vals.view(torch.int64), gidx], dim=1).contiguous()     # [S, 2] int64: value bits, global index
    if world == 1:
        h = rec.cpu().numpy().reshape(1, S, 2)
    else:
        import torch.distributed as dist

        out = torch.empty(world * S * 2, dtype=torch.int64, device=rec.device)
        dist.all_gather_into_tensor(out, rec.view(-1))
        h = out.cpu().numpy().reshape(world, S, 2)
    v = h[:, :, 0].copy().view(np.float64)
    best_v = np.empty(S)
    best_i = np.empty(S, dtype=np.int64)
    for s_ in range(S):
        best_v[s_], best_i[s_] = reduce_best(v[:, s_], h[:, s_, 1])
    return best_v, best_i
