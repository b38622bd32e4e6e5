"""Config 4's gradient exchange at its REAL size on one GPU: the flat gradient buffer of a 512^3 scene - sigma_t (V floats)
+ albedo (3V) = 2 GiB - through the one-collective all-reduce of distributed.py on the `nccl` (= RCCL) backend in a world of one rank:
gradient_support (support-mask kernels), drt_grad_pack, the collective, drt_grad_unpack, each timed with events on the launch stream.  With one
rank the sum is the input itself: the buffer must come back bit for bit.  Prints one JSON line (tools/gpu/run.sh config4ar copies it)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import uivr_amd as u
from uivr_amd import distributed as D, synthetic
from uivr_amd._native import native


def timed(fn, reps=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    res = int(os.environ.get("DRT_CONFIG4_RES", "512"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    scene = synthetic.dust_devil_scene(res=res, film=64, device=dev)
    grads = u.alloc_grads(scene)
    flat = grads["_flat"]
    B = D.COMPACT_BLOCK_FLOATS
    out = {"grid": f"{res}^3", "flat_MiB": round(flat.numel() * 4 / 2 ** 20, 1)}
    out["gradient_support_ms"] = round(timed(lambda: u.gradient_support(scene.medium.sigma_t, grads)), 3)
    sup = u.gradient_support(scene.medium.sigma_t, grads)
    n_blocks = sup.mask.numel()
    out["support_fraction"] = round(sup.count / n_blocks, 4)
    # a gradient that lives inside the support: the sigma_t part dense, the albedo part where the mask allows
    torch.manual_seed(3)
    body = flat[:n_blocks * B].view(n_blocks, B)
    body.copy_(torch.randn(1, B, device=dev).expand(n_blocks, B))
    body.mul_(sup.mask[:, None].to(torch.float32))
    ref = flat.clone()
    packed = torch.empty(sup.count * B + 1, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    out["pack_ms"] = round(timed(lambda: native().grad_pack(st, flat.data_ptr(), sup.pos.data_ptr(), n_blocks, B, packed.data_ptr(),
                                                            packed.data_ptr() + 4 * (packed.numel() - 1))), 3)
    out["unpack_ms"] = round(timed(lambda: native().grad_unpack(st, packed.data_ptr(), sup.pos.data_ptr(), n_blocks, B, flat.data_ptr())), 3)
    assert torch.equal(flat, ref), "pack -> unpack is not the identity on the support"
    out["packed_MiB"] = round(packed.numel() * 4 / 2 ** 20, 1)
    out["rccl_allreduce_packed_world1_ms"] = round(timed(lambda: dist.all_reduce(packed)), 3)
    stats = {}

    def one_collective():
        D._allreduce_flat(flat, None, "auto", stats, sup)

    out["allreduce_gradients_world1_ms"] = round(timed(one_collective, reps=3), 3)
    D.verify_pending()
    assert torch.equal(flat, ref), "the one-collective all-reduce changed a one-rank buffer"
    out["mode"], out["collectives"] = stats.get("mode"), stats.get("collectives")
    assert stats.get("collectives") == 1
    dense = {}
    out["allreduce_dense_world1_ms"] = round(timed(lambda: D._allreduce_flat(flat, None, "never", dense), reps=3), 3)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out))
    print("CONFIG4_ALLREDUCE_OK")


if __name__ == "__main__":
    main()
