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


def gather_batch(local: torch.Tensor, total: int, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All-gather per-rank shards (dim 0, sizes = shard_sizes(total, world)) into the full [total, ...] tensor.
    `out` (equal shards only): a caller-owned [total, ...] buffer to gather into, reused across calls."""
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
        shape = (total,) + tuple(local.shape[1:])
        if out is None or tuple(out.shape) != shape or out.dtype != local.dtype or out.device != local.device:
            out = torch.empty(shape, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(sizes)  # ragged: pad every shard to the largest, gather, drop the padding
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def generate_overlapped(local_generate: Callable, sem_local: torch.Tensor, x_local: torch.Tensor, num_steps: int, total: int,
                        micro_batches: int, group=None) -> torch.Tensor:
    """Local sampler in `micro_batches` slices, the all-gather of slice i running while slice i+1 computes.

    Every rank must hold an equal shard whose size is divisible by `micro_batches`.  Slice i of rank r lands at rows
    [r*Bl + i*Bm, r*Bl + (i+1)*Bm) of the result, i.e. the output is identical to the un-overlapped gather.  With the nccl
    backend each all_gather is asynchronous on the process group's stream (it waits for the slice that was just enqueued and
    overlaps the next one); the results are joined before returning.  OPT-IN: the one-shot gather stays the default until the
    overlap has been measured on a multi-GPU node."""
    world = dist.get_world_size(group)
    Bl = sem_local.shape[0]
    if Bl * world != total or micro_batches < 1 or Bl % micro_batches:
        raise ValueError(f"equal shards divisible by micro_batches required: total={total} world={world} local={Bl} micro_batches={micro_batches}")
    Bm = Bl // micro_batches
    staged = sem_local.is_cuda and dist.get_backend(group) == "gloo"  # rehearsal: gloo moves host memory
    out, works, keep = None, [], []
    for i in range(micro_batches):
        mel = local_generate(sem_local[i * Bm:(i + 1) * Bm].contiguous(), num_steps, x_local[i * Bm:(i + 1) * Bm].contiguous())
        if out is None:
            out = torch.empty((total,) + tuple(mel.shape[1:]), dtype=mel.dtype, device="cpu" if staged else mel.device)
        src = mel.cpu() if staged else mel.contiguous()
        dst = [out[r * Bl + i * Bm: r * Bl + (i + 1) * Bm] for r in range(world)]
        works.append(dist.all_gather(dst, src, group=group, async_op=True))
        keep.append(src)
    for w in works:
        w.wait()
    return out.to(sem_local.device) if staged else out


class _LocalRows:
    """Rows [lo, hi) of a [B, ...] tensor that only exists shard-wise: slicing it with exactly [lo:hi] returns the shard."""

    def __init__(self, rows: torch.Tensor, lo: int, hi: int, total: int):
        self.rows, self.lo, self.hi = rows, lo, hi
        self.shape = (total,) + tuple(rows.shape[1:])

    def __getitem__(self, sl):
        if not isinstance(sl, slice) or (sl.start, sl.stop, sl.step) != (self.lo, self.hi, None):
            raise IndexError("only the owning rank's [lo:hi] block of the start noise exists")
        return self.rows

    def new_empty(self, shape):
        return self.rows.new_empty(shape)


class ShardedEdgeInference:
    """generate_mel over a process group: every rank passes the SAME global sem_idx (and optionally the same global
    x_T); each computes its contiguous block with `local_generate` and receives the full [B, 2S, n_mels] result.

    `local_generate(sem_idx_local, num_steps, x_T_local) -> mel_local` defaults to EdgeInference.generate_mel of the
    wrapped object.  The start noise depends only on (`seed`, global utterance index), so the result does not depend on the
    number of ranks (bitwise): on the GPU each rank draws ITS rows from the library's counter-based Philox stream at its global
    offset (edtts_randn -- no rank materialises the global noise); CPU tensors (host-logic tests with a stand-in sampler) draw the
    global tensor from a torch generator and slice it."""

    def __init__(self, infer=None, local_generate: Optional[Callable] = None, group=None, micro_batches: int = 1):
        if infer is None and local_generate is None:
            raise ValueError("need an EdgeInference or a local_generate callable")
        self.infer = infer
        self.group = group
        self.micro_batches = int(micro_batches)  # > 1: overlap the all-gather with compute (generate_overlapped; opt-in)
        self._local = local_generate or (lambda sem, n, x: infer.generate_mel(sem, n, x_T=x))

    def generate_mel(self, sem_idx: torch.Tensor, num_steps: int = 4, temperature: float = 1.0, *, x_T: Optional[torch.Tensor] = None,
                     seed: int = 0, n_mels: Optional[int] = None) -> torch.Tensor:
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        B, S = sem_idx.shape
        lo, hi = shard_bounds(B, world, rank)
        if x_T is None:
            m = n_mels if n_mels is not None else self.infer.cfg.n_mels
            if sem_idx.is_cuda:
                from . import native
                if hi > lo:
                    x_loc = native.randn((hi - lo, 2 * S, m), sem_idx.device, seed, 0, lo * 2 * S * m, temperature)
                else:  # this rank owns no utterance: no draw, an empty shard for the gather
                    x_loc = torch.empty((0, 2 * S, m), dtype=torch.float32, device=sem_idx.device)
                x_T = _LocalRows(x_loc, lo, hi, B)
            else:
                g = torch.Generator(device=sem_idx.device).manual_seed(seed)
                x_T = torch.randn(B, 2 * S, m, device=sem_idx.device, generator=g) * temperature
        if self.micro_batches > 1 and B % world == 0 and (B // world) % self.micro_batches == 0:
            return generate_overlapped(self._local, sem_idx[lo:hi], x_T[lo:hi], num_steps, B, self.micro_batches, self.group)
        if hi == lo:
            # fewer utterances than ranks (e.g. BASELINE config 1, B = 1): this rank has nothing to sample -- the kernels
            # reject B < 1 -- but must still take part in the collective, with an empty shard
            local = x_T.new_empty((0,) + tuple(x_T.shape[1:]))
        else:
            local = self._local(sem_idx[lo:hi].contiguous(), num_steps, x_T[lo:hi].contiguous())
        return gather_batch(local, B, self.group)
