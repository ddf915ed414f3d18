"""Batch sharding of the noise path across the GPUs of one node.

The path shards over the batch dimension with NO data-path collective: every latent's draws are keyed by its
GLOBAL index (``noise_generation.shard_offset``), lattice/pyramid scalars that the reference shares across the
batch are keyed without a batch offset, so rank r generating latents [start, start+count) produces exactly the
slice of what one GPU would produce for the whole batch.  One process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI); collectives appear only in two OPTIONAL places:
  * ``allreduce_stats`` / ``normalise_global_``: 3 x fp64 (sum, sumsq, n) all-reduce per normalisation point, for
    exact whole-batch ``scale_noise`` parity (default semantics = every shard normalises itself, i.e. the reference
    called with B/N latents);
  * ``gather_batch``: final all-gather of the shards (the north_star's optional gather).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from . import hip_lib
from .py import noise_generation


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: (first latent, count) of rank `rank`; the first `global_batch % world` ranks get one extra."""
    if world <= 0 or not 0 <= rank < world or global_batch < 0:
        raise ValueError(f"bad shard request: batch={global_batch} rank={rank} world={world}")
    base, extra = divmod(global_batch, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def rank_world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class ShardedNoiseSampler:
    """Wraps ``make_noise_sampler`` of any noise item / chain so that each rank generates only its shard.

    ``make(x_shard, ...)`` is called with a latent of the SHARD's batch size; calls run inside
    ``shard_offset(start)`` so on-device draws use global latent indices.  (``cpu=True`` replay draws come from the
    host generator and are not shard-invariant — same as the reference run on a smaller batch.)"""

    def __init__(self, make: Callable, x_global_shape, device, *, group=None, **make_kwargs):
        self.rank, self.world = rank_world(group)
        self.group = group
        self.start, self.count = shard_range(int(x_global_shape[0]), self.rank, self.world)
        self.global_shape = tuple(x_global_shape)
        x = torch.zeros((self.count, *x_global_shape[1:]), device=device)
        with noise_generation.shard_offset(self.start):
            self.sampler = make(x, **make_kwargs)

    def __call__(self, sigma, sigma_next) -> torch.Tensor:
        with noise_generation.shard_offset(self.start):
            return self.sampler(sigma, sigma_next)

    def gather(self, local: torch.Tensor, *, direct: bool = False) -> torch.Tensor:
        return gather_batch(local, self.global_shape[0], group=self.group, direct=direct)


def _host_staged(t: torch.Tensor, group=None) -> bool:
    """A gloo process group (ranks sharing one GPU in a self-test, no RCCL) moves device tensors through the host."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def allreduce_stats(local: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the (sum, sumsq, n) fp64 triple over ranks (24 bytes: latency only)."""
    if local.dtype != torch.float64 or local.numel() != 3:
        raise ValueError("allreduce_stats expects a float64 tensor of 3 elements")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if _host_staged(local, group):
            host = local.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            local.copy_(host)
        else:
            dist.all_reduce(local, op=dist.ReduceOp.SUM, group=group)
    return local


def normalise_global_(x: torch.Tensor, factor: float = 1.0, *, threshold_std_devs: float = 2.5, group=None) -> torch.Tensor:
    """scale_noise(x, factor, normalized=True) with statistics of the WHOLE logical batch (all ranks): HIP stats kernel ->
    3-double all-reduce -> HIP apply kernel with the global element count.  No host sync."""
    total = allreduce_stats(hip_lib.stats_finalize(hip_lib.stats(x), x.numel()), group)
    partials = torch.zeros(hip_lib.NPART * 2, dtype=torch.float64, device=x.device)
    partials[:2] = total[:2]
    n_total = int(total[2].item())  # optional exact-parity path: one 8-byte read-back
    return hip_lib.scale_noise_(x, factor, True, partials, threshold_std_devs=threshold_std_devs, n_total=n_total)


def gather_batch(local: torch.Tensor, global_batch: int, group=None, *, direct: bool = False) -> torch.Tensor:
    """All-gather the shards along dim 0 into [global_batch, ...] on every rank.

    ``direct=False``: one RCCL ``all_gather_into_tensor`` (uneven shards are padded to the largest).  On MI355X the xGMI fabric is
    point-to-point (7 links x ~153 GB/s per GPU), so a ring all-gather is bound by ONE link per hop.
    ``direct=True``: every rank posts world - 1 sends of its shard and world - 1 receives straight into the slices
    of the output, all at once (``batch_isend_irecv`` -> RCCL point-to-point): every link carries one shard concurrently, nothing
    is forwarded, uneven shards need no padding.  Either way the gather is outside the timed hot path unless the consumer needs the
    full batch on one device."""
    rank, world = rank_world(group)
    if world == 1:
        return local
    spans = [shard_range(global_batch, r, world) for r in range(world)]
    counts = [c for _s, c in spans]
    if local.shape[0] != counts[rank]:
        raise ValueError(f"gather_batch: rank {rank} holds {local.shape[0]} latents, its shard of {global_batch} is {counts[rank]}")
    local = local.contiguous()
    if _host_staged(local, group):
        return gather_batch(local.cpu(), global_batch, group, direct=direct).to(local.device)
    if direct:
        out = torch.empty((global_batch, *local.shape[1:]), dtype=local.dtype, device=local.device)
        start, count = spans[rank]
        out[start:start + count].copy_(local)
        ops = []
        for step in range(1, world):  # peer order rotated by rank: at every step each link pair is used once
            dst, src = (rank + step) % world, (rank - step) % world
            if count:
                ops.append(dist.P2POp(dist.isend, local, dst if group is None else dist.get_global_rank(group, dst), group))
            s0, c0 = spans[src]
            if c0:
                ops.append(dist.P2POp(dist.irecv, out[s0:s0 + c0], src if group is None else dist.get_global_rank(group, src), group))
        for req in dist.batch_isend_irecv(ops) if ops else ():
            req.wait()
        return out
    biggest = max(counts)
    send = local
    if local.shape[0] < biggest:
        pad = torch.zeros((biggest - local.shape[0], *local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat((local, pad))
    out = torch.empty((world * biggest, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    if all(c == biggest for c in counts):
        return out
    return torch.cat([out[r * biggest: r * biggest + counts[r]] for r in range(world)])
