"""Latent sharding over the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl"
is RCCL on ROCm; "gloo" for the CPU rehearsal of the host logic).

After the OILMM projection the L latent processes are independent (reference moihgp.h:217-221 is a plain
loop with no cross-latent term), so rank r owns the contiguous latents [lo_r, hi_r), runs its recursion with
no data-path collective, and only the scalar sum of per-latent NLLs -- the quantity L-BFGS consumes
(moihgp.h:684 `loss += ...`) -- is all-reduced: 8 bytes over xGMI, pure latency.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_bounds(L: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced partition of L latents; the first L % world_size ranks get one extra."""
    q, r = divmod(L, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def allreduce_nll(nll_local: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of the per-latent NLLs over all shards: one fp64 scalar all-reduce (SUM)."""
    total = nll_local.sum(dtype=torch.float64).reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if total.is_cuda and dist.get_backend(group) == "gloo":      # CPU rehearsal of the exchange
            t = total.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return t.to(total.device)
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return total


def gather_latent_grads(grad_local: torch.Tensor, L: int, group=None) -> torch.Tensor:
    """Per-latent gradients are disjoint across shards: all-gather [L_r, P] blocks into [L, P]
    (mode G only; moihgp.h:608-609 packs them latent-major)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grad_local
    ws = dist.get_world_size(group)
    P = grad_local.shape[1]
    sizes = [shard_bounds(L, ws, r)[1] - shard_bounds(L, ws, r)[0] for r in range(ws)]
    mx = max(sizes)
    pad = torch.zeros((mx, P), dtype=grad_local.dtype, device=grad_local.device)
    pad[: grad_local.shape[0]] = grad_local
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)
