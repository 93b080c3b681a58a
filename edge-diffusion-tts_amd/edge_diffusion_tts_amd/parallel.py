"""Batch sharding of the sampler over the GPUs of one node (new functionality; the reference is single-device).

Utterances are independent on this path (no cross-batch operation anywhere: norms are per frame, attention per
utterance), so the batch is partitioned into contiguous blocks, one per rank, with no data-path collective; the only
exchange is ONE all-gather of the final mel shard per call (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm).
Weights are replicated (7.9 MB), every rank packs its own copy.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; the first total % world ranks get one extra utterance."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_sizes(total: int, world: int) -> List[int]:
    return [hi - lo for lo, hi in (shard_bounds(total, world, r) for r in range(world))]


def gather_batch(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All-gather per-rank shards (dim 0, sizes = shard_sizes(total, world)) into the full [total, ...] tensor."""
    world = dist.get_world_size(group)
    sizes = shard_sizes(total, world)
    rank = dist.get_rank(group)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: local shard has {local.shape[0]} rows, expected {sizes[rank]}")
    local = local.contiguous()
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (e.g. several ranks sharing one GPU): gloo moves host memory, so stage through the CPU
        return gather_batch(local.cpu(), total, group).to(local.device)
    if len(set(sizes)) == 1:  # the common case: one in-place collective into the output buffer
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(sizes)  # ragged: pad every shard to the largest, gather, drop the padding
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


class ShardedEdgeInference:
    """generate_mel over a process group: every rank passes the SAME global sem_idx (and optionally the same global
    x_T); each computes its contiguous block with `local_generate` and receives the full [B, 2S, n_mels] result.

    `local_generate(sem_idx_local, num_steps, x_T_local) -> mel_local` defaults to EdgeInference.generate_mel of the
    wrapped object.  The start noise is drawn for the GLOBAL batch from `seed` and sliced, so the result does not depend
    on the number of ranks (bitwise)."""

    def __init__(self, infer=None, local_generate: Optional[Callable] = None, group=None):
        if infer is None and local_generate is None:
            raise ValueError("need an EdgeInference or a local_generate callable")
        self.infer = infer
        self.group = group
        self._local = local_generate or (lambda sem, n, x: infer.generate_mel(sem, n, x_T=x))

    def generate_mel(self, sem_idx: torch.Tensor, num_steps: int = 4, temperature: float = 1.0, *, x_T: Optional[torch.Tensor] = None,
                     seed: int = 0, n_mels: Optional[int] = None) -> torch.Tensor:
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        B, S = sem_idx.shape
        lo, hi = shard_bounds(B, world, rank)
        if x_T is None:
            m = n_mels if n_mels is not None else self.infer.cfg.n_mels
            g = torch.Generator(device=sem_idx.device).manual_seed(seed)
            x_T = torch.randn(B, 2 * S, m, device=sem_idx.device, generator=g) * temperature
        local = self._local(sem_idx[lo:hi].contiguous(), num_steps, x_T[lo:hi].contiguous())
        return gather_batch(local, B, self.group)
