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
COMPACT_MAX_ACTIVE = 0.7            # above this fraction of blocks in the packing set the dense all-reduce is cheaper
HISTORY_RESET_CALLS = 256           # a packing set grown from the gradients seen so far is rebuilt after this many calls


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


_PINNED = {"buf": None, "next": 0}


def _pinned_slot(dtype=torch.int32) -> torch.Tensor:
    """One 4-byte word (int32 or float32) of a small ring of pinned host memory (a slot is read long before the ring comes
    round again; a pinned allocation per step would cost more than the kernels it serves)."""
    if _PINNED["buf"] is None:
        _PINNED["buf"] = torch.empty(256, dtype=torch.int32, pin_memory=True)
    i = _PINNED["next"]
    _PINNED["next"] = (i + 1) % 256
    slot = _PINNED["buf"][i:i + 1]
    return slot if dtype == torch.int32 else slot.view(dtype)


def _positions(mask: torch.Tensor):
    """(pos, count): pos[b] = rank of block b among the blocks of the set `mask` (uint8), -1 outside it; count = their number, a
    one-element int32 tensor on the mask's device.  Device masks: drt_grad_block_positions (two small kernels)."""
    n = mask.numel()
    if mask.is_cuda:
        from ._native import native
        pos = torch.empty(n, dtype=torch.int32, device=mask.device)
        cnt = torch.empty(1, dtype=torch.int32, device=mask.device)
        scratch = torch.empty((n + 1023) // 1024 + 1, dtype=torch.int32, device=mask.device)
        with torch.cuda.device(mask.device):
            native().grad_block_positions(torch.cuda.current_stream().cuda_stream, mask.data_ptr(), n, pos.data_ptr(), cnt.data_ptr(),
                                          scratch.data_ptr())
        return pos, cnt
    m = mask.to(torch.int32)
    pos = torch.cumsum(m, dim=0, dtype=torch.int32) - 1
    pos = torch.where(m != 0, pos, torch.full_like(pos, -1))
    return pos, m.sum(dtype=torch.int32).reshape(1)


class GradientSupport:
    """A set of 256-byte blocks of the flat gradient buffer OUTSIDE of which the gradient of every rank is zero - known
    before the backward pass, the same on every rank (it is computed from the replicated parameters, no communication).

    `gradient_support(...)` builds it at the start of a step; the number of blocks travels to the host behind the render
    passes (asynchronous copy + event), so that the all-reduce at the end of the backward can size its packed buffer
    without stalling the device."""

    def __init__(self, mask: torch.Tensor, n_floats: int):
        self.mask = mask.to(torch.uint8).contiguous()       # [n_blocks], 1 = the block may be non-zero
        self.n_floats = int(n_floats)
        self._count = None
        self._host, self._ready = None, None
        self.pos, cnt = _positions(self.mask)               # where each block of the set goes in the packed buffer
        if self.mask.is_cuda:
            with torch.cuda.device(self.mask.device):
                self._host = _pinned_slot()                          # (a pinned allocation per step would cost more than the kernels)
                self._host.copy_(cnt, non_blocking=True)
                self._ready = torch.cuda.Event()
                self._ready.record(torch.cuda.current_stream(self.mask.device))
        else:
            self._count = int(cnt)

    @property
    def count(self) -> int:
        if self._count is None:
            self._ready.synchronize()                       # (recorded a render pass ago: does not wait in a training loop)
            self._count = int(self._host[0])
        return self._count


def gradient_support(sigma_t: torch.Tensor, grads: Dict[str, torch.Tensor], sparse_keys=("medium1.albedo.data",),
                     block_floats: int = COMPACT_BLOCK_FLOATS) -> Optional[GradientSupport]:
    """Where can the gradient grids `grads` (render.alloc_grads: views of `grads["_flat"]`) be non-zero, given sigma_t?

    * `sparse_keys` - per-voxel grids that only receive splats at REAL collisions (albedo: python/integrators/
      volpathsimple.py:152-172 splats at a scattering vertex, whose sigma_t(x) > 0; the DRT vertex's albedo gradient
      `:577-581` carries the factor sigma_t(x')): a voxel is reached from a point whose interpolated sigma_t is non-zero,
      i.e. whose 8 surrounding voxels are not all zero - the voxels within one step (3x3x3 neighbourhood) of a non-zero one;
    * every other grid (sigma_t: the four transmittance samples of backpropagate_transmittance land anywhere on a segment,
      `:584-607`): every block.
    None where that reasoning does not apply as is (no flat buffer, a sparse grid of another resolution)."""
    flat = grads.get("_flat")
    if flat is None or flat.numel() < 64 * block_floats:
        return None
    n_floats = flat.numel()
    n_blocks = n_floats // block_floats
    if sigma_t.device != flat.device:                               # (a replica on another device: the mask belongs next to the buffer)
        sigma_t = sigma_t.to(flat.device)
    # device buffers with ONE per-voxel plane (volpathsimple's albedo): two small kernels (a non-zero bit per voxel, then one thread
    # per block: drt_grad_support_mask, 0.05 ms at 256^3 - the torch formulation below takes 0.86 ms, a fifth of a rank's step at 8 GPUs)
    sparse = [(k, g) for k, g in grads.items() if k != "_flat" and k in sparse_keys]
    if (flat.is_cuda and len(sparse) == 1 and sigma_t.is_cuda and sigma_t.dtype == torch.float32 and sigma_t.is_contiguous()
            and flat.dtype == torch.float32 and tuple(sparse[0][1].shape[:3]) == tuple(sigma_t.shape[:3])):
        g = sparse[0][1]
        off = (g.data_ptr() - flat.data_ptr()) // flat.element_size()
        if 0 <= off and off + g.numel() <= n_floats and g.is_contiguous():
            from ._native import native
            rz, ry, rx = (int(v) for v in sigma_t.shape[:3])
            ch = g.numel() // (rx * ry * rz)
            mask = torch.empty(n_blocks, dtype=torch.uint8, device=flat.device)
            bits = torch.empty(((rx + 31) // 32) * ry * rz, dtype=torch.int32, device=flat.device)
            with torch.cuda.device(flat.device):
                native().grad_support_mask(torch.cuda.current_stream().cuda_stream, sigma_t.data_ptr(), rx, ry, rz, off, ch, n_blocks,
                                           block_floats, bits.data_ptr(), mask.data_ptr())
            return GradientSupport(mask, n_floats)
    may = torch.zeros(n_blocks * block_floats, dtype=torch.bool, device=flat.device)
    occ = None
    for k, g in grads.items():
        if k == "_flat":
            continue
        off = (g.data_ptr() - flat.data_ptr()) // flat.element_size()
        lo, hi = off, min(off + g.numel(), n_blocks * block_floats)
        if not (0 <= off and off + g.numel() <= n_floats):
            return None                                             # not a view of the flat buffer
        if hi <= lo:
            continue
        if k in sparse_keys:
            if tuple(g.shape[:3]) != tuple(sigma_t.shape[:3]):
                return None
            if occ is None:
                o = (sigma_t.reshape(sigma_t.shape[:3]) != 0).to(torch.float32)[None, None]
                occ = torch.nn.functional.max_pool3d(o, kernel_size=3, stride=1, padding=1)[0, 0].reshape(-1) > 0     # [V]
            ch = g.numel() // occ.numel()
            may[lo:hi] = occ.repeat_interleave(ch)[:hi - lo]
        else:
            may[lo:hi] = True
    return GradientSupport(may.view(n_blocks, block_floats).any(dim=1), n_floats)


class _ReduceState:
    """What one (process group, buffer) remembers between backward passes."""
    __slots__ = ("in_set", "src", "pos", "count", "calls", "pending", "dev_mode")

    def __init__(self):
        self.in_set = None          # uint8 [n_blocks]: the packing set grown from the gradients seen so far
        self.src = None             # int64 [count]: its block indices (host tensors: the torch formulation of the packing)
        self.pos = None             # int32 [n_blocks]: rank of every block in the set, -1 outside (device buffers: the packing kernels)
        self.count = 0
        self.calls = 0
        self.pending = None         # (pinned host float, event): blocks found outside a GradientSupport, not looked at yet
        self.dev_mode = None        # which of src / pos the history set was laid out for (the packing kernels or the torch formulation)


_STATE: Dict[tuple, _ReduceState] = {}


def _state_for(flat: torch.Tensor, group) -> _ReduceState:
    key = (id(group) if group is not None else None, str(flat.device), flat.numel())
    st = _STATE.get(key)
    if st is None:
        st = _STATE[key] = _ReduceState()
    return st


def reset_allreduce_state() -> None:
    """Forget every packing set (a new process group, a new scene)."""
    verify_pending()
    _STATE.clear()


def verify_pending() -> None:
    """The all-reduce that packs with a `GradientSupport` does not wait for its own check (the number of non-zero blocks
    it found outside the support, summed over the ranks, travels to the host behind the collective); the next call, or
    this function at the end of a run, looks at it.  A non-zero count means the support was not one: that step's
    gradient is missing the other ranks' share of those blocks - raised, never silent."""
    for st in _STATE.values():
        if st.pending is not None:
            host, ev = st.pending
            st.pending = None
            ev.synchronize()
            if float(host[0]) != 0.0:
                raise RuntimeError(f"allreduce_gradients: {float(host[0]):.0f} non-zero gradient blocks lay outside the "
                                   "GradientSupport of an earlier backward pass (its all-reduce left them unsummed); "
                                   "pass support=None or strict=True")


def _pack(body: torch.Tensor, tail: torch.Tensor, src: torch.Tensor, extra: torch.Tensor) -> torch.Tensor:
    return torch.cat([body.index_select(0, src).reshape(-1), tail, extra])


def _allreduce_flat(flat: torch.Tensor, group, compact, stats: Optional[dict], support: Optional[GradientSupport] = None,
                    strict: Optional[bool] = None) -> None:
    """Sum `flat` over the group in place.  `compact`: "auto" (default) | "never" | "always".

    The gradient of a sparse volume is mostly exact zeros: the albedo planes (3/4 of the buffer) only receive splats
    where real scattering happens, i.e. next to voxels with sigma_t > 0.  So only a SET of 256-byte blocks is packed
    and summed - ONE collective per backward - and the packed buffer carries, as its last float, the number of non-zero
    blocks this rank holds OUTSIDE the set: a non-zero sum says the set was too small.  Blocks outside the set are zero
    on every rank otherwise, so the result equals the dense sum (up to the summation order inside the collective);
    non-finite values count as non-zero and propagate.  Where the set comes from:

    * `support` (a `GradientSupport`, from the replicated parameters: rigorous, known before the pass): one collective and
      no host wait - the check is looked at by the NEXT call (`verify_pending`), or right away with `strict=True` / on
      host tensors, where a violated support falls back to the dense collective;
    * no `support`: the union of the non-zero blocks of the sums seen so far for this (group, buffer).  One collective;
      the host looks at the check before the result is scattered (a set grown from history can be outgrown: then the
      dense collective runs as well and the set grows).  The first call, and every HISTORY_RESET_CALLS-th, agrees on
      the set with an all-reduce(MAX) of one byte per block first.
    Every decision is made from values all ranks hold identically (the agreed set, its size, the summed check)."""
    import torch.distributed as dist
    n = flat.numel()
    B = COMPACT_BLOCK_FLOATS
    if compact == "never" or n < 64 * B:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if stats is not None:
            stats.update(mode="dense", floats=n, active_fraction=1.0, collectives=1)
        return
    import contextlib
    with (torch.cuda.device(flat.device) if flat.is_cuda else contextlib.nullcontext()):
        _allreduce_flat_body(flat, group, compact, stats, support, strict, dist, n, B)


def _allreduce_flat_body(flat, group, compact, stats, support, strict, dist, n, B) -> None:
    n_full = (n // B) * B
    body = flat[:n_full].view(-1, B)
    tail = flat[n_full:]
    n_blocks = body.shape[0]
    st = _state_for(flat, group)
    verify_pending()
    st.calls += 1
    n_coll = 0

    def dense(frac):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if stats is not None:
            stats.update(mode="dense", floats=n, active_fraction=frac, collectives=n_coll + 1)

    if support is not None and (support.n_floats != n or support.mask.numel() != n_blocks or support.mask.device != flat.device):
        raise ValueError("allreduce_gradients: `support` was built for another buffer")

    # ---- the packing set ------------------------------------------------------------------------------------------------
    dev_path = (flat.is_cuda and flat.dtype == torch.float32 and flat.data_ptr() % 16 == 0        # the packing kernels ...
                and B in (64, 128, 256))                                                         # ... are instantiated for these block sizes
    if support is not None:
        in_set, count, pos = support.mask, support.count, support.pos
        src = None
    else:
        if st.in_set is None or st.calls % HISTORY_RESET_CALLS == 0:
            mask = _block_mask(body)                               # agree on the set: one byte per block, MAX
            dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=group)
            n_coll += 1
            _set_history(st, mask, dev_path)                       # (host wait: first call and resets only)
        elif st.dev_mode != dev_path:
            # the same (group, buffer size) came back with another alignment / dtype: the set stays, its index tables are rebuilt for
            # the formulation that runs now (pos for the packing kernels, src for the torch one) - never a None table, never a
            # nonzero() per call
            _set_history(st, st.in_set, dev_path)
        in_set, count, src, pos = st.in_set, st.count, st.src, st.pos
    frac = count / n_blocks
    if compact != "always" and frac > COMPACT_MAX_ACTIVE:          # (known before anything is packed)
        return dense(frac)

    # ---- one collective: the set's blocks + the ragged tail + the check ---------------------------------------------------
    if count == 0 and tail.numel() == 0 and support is None and n_coll:
        if stats is not None:                                      # (the mask collective just said: zeros on every rank)
            stats.update(mode="compact", floats=0, active_fraction=0.0, collectives=n_coll)
        return
    if dev_path:
        # one pass over the buffer: the set's blocks to their places, the others tested for the check (drt_grad_pack)
        from ._native import native
        packed = torch.empty(count * B + tail.numel() + 1, dtype=torch.float32, device=flat.device)
        packed[count * B:].zero_()
        if tail.numel():
            packed[count * B:count * B + tail.numel()].copy_(tail)
        native().grad_pack(torch.cuda.current_stream().cuda_stream, flat.data_ptr(), pos.data_ptr(), n_blocks, B, packed.data_ptr(),
                           packed.data_ptr() + 4 * (packed.numel() - 1))
    else:
        if src is None:
            src = torch.nonzero(in_set).reshape(-1)
        local = _block_mask(body)
        outside = (local & (1 - in_set)).sum(dtype=torch.float32).reshape(1)
        packed = _pack(body, tail, src, outside)
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    n_coll += 1
    lazy = support is not None and flat.is_cuda and not strict
    if lazy:
        host = _pinned_slot(torch.float32)
        host.copy_(packed[-1:], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(flat.device))
        st.pending = (host, ev)
        missed = 0.0
    else:
        missed = float(packed[-1])                                 # the host waits for the collective here
    if missed != 0.0:
        # the set was outgrown: nothing has been scattered yet, `flat` still holds this rank's gradient
        dense(frac)
        if support is None:                                        # grow the set by what the dense sum shows
            _set_history(st, torch.maximum(st.in_set, _block_mask(body)), dev_path)
        if stats is not None:
            stats.update(outgrown=True)
        return
    if dev_path:
        native().grad_unpack(torch.cuda.current_stream().cuda_stream, packed.data_ptr(), pos.data_ptr(), n_blocks, B, flat.data_ptr())
    else:
        body.index_copy_(0, src, packed[:count * B].view(-1, B))
    if tail.numel():
        tail.copy_(packed[count * B:count * B + tail.numel()])
    if stats is not None:
        stats.update(mode="compact", floats=count * B + tail.numel(), sent_floats=packed.numel(), active_fraction=frac,
                     collectives=n_coll)


def _set_history(st: _ReduceState, mask: torch.Tensor, dev_path: bool) -> None:
    """The packing set grown from the sums seen so far (no `GradientSupport`): its size is needed on the host."""
    st.in_set = mask
    st.dev_mode = dev_path
    if dev_path:
        st.pos, cnt = _positions(mask)
        st.src = None
        st.count = int(cnt)
    else:
        st.src = torch.nonzero(mask).reshape(-1)
        st.pos = None
        st.count = int(st.src.numel())


def allreduce_gradients(grads: Dict[str, torch.Tensor], group=None, shard: Optional[ShardSpec] = None,
                        compact: str = "auto", stats: Optional[dict] = None, support: Optional[GradientSupport] = None,
                        strict: Optional[bool] = None) -> None:
    """Sum the gradient grids over all ranks, in place: ONE collective per backward (see `_allreduce_flat`).  The grids
    live in (or are flattened into) ONE buffer (sigma_t: V floats + albedo: 3V floats) so that a single large all-reduce
    crosses xGMI instead of one per parameter.

    Only PARTITIONED work is summed: with `shard` given, the call is a no-op unless `shard.world > 1`
    (every rank of an unsharded render computed the full gradient already; summing those would
    multiply it by the world size).  `stats`, if given, receives {"mode", "floats", "active_fraction", "collectives"}.
    `support`: the blocks that can be non-zero (`gradient_support`); only used with a `_flat` buffer."""
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
        _allreduce_flat(grads["_flat"], group, compact, stats, support, strict)
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
