"""Image-tile data parallelism for the DRT path.

The reference is single-GPU (no collective anywhere).  Rays are independent given
the read-only grids; the only cross-ray state is the additive gradient grid.  So:
one process per GPU, replicated parameter grids, the image's pixels dealt out in
interleaved chunks (chunk c -> rank c % world, which balances empty-space tiles),
random streams keyed by the GLOBAL ray index so that the result does not depend on
the number of ranks (up to fp summation order), and ONE all-reduce (RCCL over xGMI
when the backend is "nccl") of the gradient grids per backward.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch


@dataclass(frozen=True)
class ShardSpec:
    """Interleaved pixel-chunk sharding of an image with `n_pixels` pixels."""
    rank: int = 0
    world: int = 1
    chunk_pixels: int = 2048

    def __post_init__(self):
        if not (0 <= self.rank < self.world):
            raise ValueError(f"rank {self.rank} outside [0, {self.world})")
        if self.chunk_pixels < 1:
            raise ValueError("chunk_pixels must be >= 1")

    def check(self, n_pixels: int) -> None:
        if self.world > 1 and n_pixels % (self.chunk_pixels * self.world) != 0:
            raise ValueError(f"n_pixels={n_pixels} must be a multiple of chunk_pixels*world="
                             f"{self.chunk_pixels * self.world} (pick chunk_pixels accordingly)")

    def n_local_pixels(self, n_pixels: int) -> int:
        self.check(n_pixels)
        return n_pixels // self.world

    def ray_mapping(self, spp: int):
        """(ray_offset, (chunk_rays, stride_rays)) for drt_set_ray_interleave."""
        if self.world == 1:
            return 0, None
        chunk = self.chunk_pixels * spp
        return self.rank * chunk, (chunk, chunk * self.world)

    def pixel_indices(self, n_pixels: int, device=None) -> torch.Tensor:
        """Global pixel index of every local pixel (for gathering reference values)."""
        n_local = self.n_local_pixels(n_pixels)
        i = torch.arange(n_local, device=device, dtype=torch.int64)
        if self.world == 1:
            return i
        return (i // self.chunk_pixels) * (self.chunk_pixels * self.world) \
            + self.rank * self.chunk_pixels + (i % self.chunk_pixels)

    def batch_range(self, batch_size: int):
        """(first, count): this rank's contiguous share of a `batch_size` list of random (sensor, pixel)
        entries (render_batch; SURVEY.md 8e "shard the batch_size pixel list").  The entries are i.i.d., so a
        contiguous split is balanced; ragged sizes are allowed (counts differ by at most one)."""
        first = (batch_size * self.rank) // self.world
        return first, (batch_size * (self.rank + 1)) // self.world - first

    @property
    def partitioned(self) -> bool:
        """True when the work of a render call was dealt across ranks (the gradients then need the all-reduce)."""
        return self.world > 1

    @staticmethod
    def default_chunk(n_pixels: int, world: int, target: int = 2048) -> int:
        """Largest chunk <= target such that n_pixels % (chunk*world) == 0."""
        c = min(target, max(1, n_pixels // world))
        while c > 1 and n_pixels % (c * world) != 0:
            c -= 1
        return c


def from_environment(n_pixels: Optional[int] = None) -> ShardSpec:
    """ShardSpec of the current torch.distributed process group (world 1 if none)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(), dist.get_rank()
    else:
        world, rank = 1, 0
    chunk = ShardSpec.default_chunk(n_pixels, world) if n_pixels else 2048
    return ShardSpec(rank=rank, world=world, chunk_pixels=chunk)


def allreduce_gradients(grads: Dict[str, torch.Tensor], group=None, shard: Optional[ShardSpec] = None) -> None:
    """Sum the gradient grids over all ranks, in place: one collective per backward.
    The grids live in (or are flattened into) ONE buffer (sigma_t: V floats + albedo:
    3V floats) so that a single large all-reduce crosses xGMI instead of one per parameter.

    Only PARTITIONED work is summed: with `shard` given, the call is a no-op unless `shard.world > 1`
    (every rank of an unsharded render computed the full gradient already; summing those would
    multiply it by the world size)."""
    import torch.distributed as dist
    if shard is not None and not shard.partitioned:
        return
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    if shard is not None and shard.world != dist.get_world_size(group):
        raise ValueError(f"ShardSpec.world={shard.world} does not match the process group size {dist.get_world_size(group)}")
    if "_flat" in grads:      # render.alloc_grads: the grids are views of one buffer
        dist.all_reduce(grads["_flat"], op=dist.ReduceOp.SUM, group=group)
        return
    keys = sorted(grads)
    flat = torch.cat([grads[k].reshape(-1) for k in keys])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for k in keys:
        n = grads[k].numel()
        grads[k].copy_(flat[off:off + n].view_as(grads[k]))
        off += n


def local_loss_scale(n_local: int, n_global: int) -> float:
    """Factor that turns a loss normalised by the LOCAL entry count (every loss of `losses.py` divides by
    `img.numel()`) into this rank's additive share of the loss over the global image / batch:
    sum_ranks scale_r * loss_r == loss(global).  Back-propagating `scale * loss_local` on every rank and
    all-reducing (SUM) the gradient grids then yields exactly the gradient of the global loss."""
    return float(n_local) / float(n_global)


def allreduce_scalar(value: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of a per-rank partial loss (separable pixel sums)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        value = value.clone()
        dist.all_reduce(value, op=dist.ReduceOp.SUM, group=group)
    return value
