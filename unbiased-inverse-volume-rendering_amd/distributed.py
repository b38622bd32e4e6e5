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


COMPACT_BLOCK_FLOATS = 64           # 256 B of the flat gradient buffer per block (64 | 128 | 256)
COMPACT_MAX_ACTIVE = 0.7            # above this fraction of non-zero blocks the dense all-reduce is cheaper


def _block_mask(body: torch.Tensor) -> torch.Tensor:
    """uint8 [n_blocks]: 1 where a row of `body` holds anything but zeros (NaN / inf count as non-zero).
    Device buffers: drt_grad_block_mask (one streaming pass at HBM rate; torch's row reductions take 3x longer)."""
    if body.is_cuda:
        from ._native import native
        mask = torch.empty(body.shape[0], dtype=torch.uint8, device=body.device)
        with torch.cuda.device(body.device):
            native().grad_block_mask(torch.cuda.current_stream().cuda_stream, body.data_ptr(), body.shape[0], body.shape[1],
                                     mask.data_ptr())
        return mask
    return (body != 0).any(dim=1).to(torch.uint8)


_PACK_CAPACITY: Dict[tuple, int] = {}     # (device, buffer size) -> blocks the packed buffer was sized for last time


def _capacity_for(count: int, n_blocks: int) -> int:
    return min(n_blocks, count + count // 4 + 64)


def _allreduce_flat(flat: torch.Tensor, group, compact, stats: Optional[dict]) -> None:
    """Sum `flat` over the group in place.  `compact`: "auto" (default) | "never" | "always".

    The gradient of a sparse volume is mostly exact zeros: the albedo planes (3/4 of the buffer) only receive
    splats where real scattering happens, i.e. next to voxels with sigma_t > 0, and with a majorant supergrid
    (majorant_resolution_factor > 0) the sigma_t plane only where the local majorant is positive.  So the ranks first
    agree on the set of 256-B blocks that are non-zero on ANY rank (an all-reduce(MAX) of one byte per block:
    1 MiB for a 256^3 medium), and, when that set is small enough to pay for the packing, all-reduce only those
    blocks.  Blocks outside the set are zero on every rank, so the sum is the same as the dense one (up to the
    summation order inside the collective); non-finite values count as non-zero and propagate.

    No pipeline stall: the packed buffer is sized from the PREVIOUS call's block count (+25 %), so the packing kernels
    are enqueued without knowing this call's count; the count travels to the host meanwhile (asynchronous copy + event)
    and is only waited for right before the collective is enqueued - the device is busy packing by then.  A count beyond
    the capacity (the block set grew by more than 25 % since the last step) falls back to the dense collective."""
    import torch.distributed as dist
    n = flat.numel()
    B = COMPACT_BLOCK_FLOATS
    if compact == "never" or n < 64 * B:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if stats is not None:
            stats.update(mode="dense", floats=n, active_fraction=1.0)
        return
    n_full = (n // B) * B
    body = flat[:n_full].view(-1, B)
    tail = flat[n_full:]
    mask = _block_mask(body)
    dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=group)
    n_blocks = mask.numel()
    cs = torch.cumsum(mask, dim=0, dtype=torch.int32)      # inclusive: cs[b] = non-zero blocks up to and including b
    key = (str(flat.device), n)
    cap = _PACK_CAPACITY.get(key)
    count_host, ready = None, None
    if flat.is_cuda and cap is not None:                   # the count goes to the host behind the packing kernels
        count_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        count_host.copy_(cs[-1:], non_blocking=True)
        ready = torch.cuda.Event()
        ready.record()
    else:                                                  # first call for this buffer (nothing to size from), or host tensors
        count = int(cs[-1])
        if cap is None:
            cap = _capacity_for(count, n_blocks)

    def pack(capacity):
        # row j of the packed buffer = the j-th non-zero block (a search in the running count), zeros past the count
        j = torch.arange(1, capacity + 1, dtype=torch.int32, device=flat.device)
        src = torch.searchsorted(cs, j).clamp_(max=n_blocks - 1)
        rows = torch.where((j <= cs[-1])[:, None], body.index_select(0, src), torch.zeros((), dtype=flat.dtype, device=flat.device))
        return src, torch.cat([rows.reshape(-1), tail])

    src, packed = pack(cap)
    if ready is not None:
        ready.synchronize()                                # (the packing above is still running or queued)
        count = int(count_host[0])
    frac = count / n_blocks
    _PACK_CAPACITY[key] = _capacity_for(count, n_blocks)
    if count == 0 and tail.numel() == 0:                   # all zeros on every rank: nothing to sum
        if stats is not None:
            stats.update(mode="compact", floats=0, active_fraction=0.0)
        return
    if count > cap and compact == "always":
        cap = count
        src, packed = pack(cap)
    if (compact != "always" and frac > COMPACT_MAX_ACTIVE) or count > cap:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if stats is not None:
            stats.update(mode="dense", floats=n, active_fraction=frac)
        return
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    body.index_copy_(0, src[:count].to(torch.int64), packed[:count * B].view(-1, B))
    if tail.numel():
        tail.copy_(packed[cap * B:])
    if stats is not None:
        stats.update(mode="compact", floats=count * B + tail.numel(), sent_floats=packed.numel(), active_fraction=frac)


def allreduce_gradients(grads: Dict[str, torch.Tensor], group=None, shard: Optional[ShardSpec] = None,
                        compact: str = "auto", stats: Optional[dict] = None) -> None:
    """Sum the gradient grids over all ranks, in place: one collective per backward (plus a one-byte-per-block
    mask, see `_allreduce_flat`).  The grids live in (or are flattened into) ONE buffer (sigma_t: V floats +
    albedo: 3V floats) so that a single large all-reduce crosses xGMI instead of one per parameter.

    Only PARTITIONED work is summed: with `shard` given, the call is a no-op unless `shard.world > 1`
    (every rank of an unsharded render computed the full gradient already; summing those would
    multiply it by the world size).  `stats`, if given, receives {"mode", "floats", "active_fraction"}."""
    import torch.distributed as dist
    if compact not in ("auto", "never", "always"):
        raise ValueError(f"compact must be 'auto', 'never' or 'always', not {compact!r}")
    if shard is not None and not shard.partitioned:
        return
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    if shard is not None and shard.world != dist.get_world_size(group):
        raise ValueError(f"ShardSpec.world={shard.world} does not match the process group size {dist.get_world_size(group)}")
    if "_flat" in grads:      # render.alloc_grads: the grids are views of one buffer
        _allreduce_flat(grads["_flat"], group, compact, stats)
        return
    keys = sorted(grads)
    flat = torch.cat([grads[k].reshape(-1) for k in keys])
    _allreduce_flat(flat, group, compact, stats)
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
