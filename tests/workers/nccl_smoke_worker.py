"""RCCL smoke: a one-rank `nccl` process group on the GPU - communicator initialisation and the
collectives of the gradient all-reduce (uint8 MAX on the block mask, float SUM on the packed / dense buffer, the scalar
loss SUM) execute at least once on the backend the 8-GPU runs use.  With one rank the sums are the inputs themselves."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import uivr_amd as u
from uivr_amd import distributed as D


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=int(os.environ.get("RANK", "0")),
                            world_size=int(os.environ.get("WORLD_SIZE", "1")), device_id=dev)
    assert dist.get_backend() == "nccl"
    B = D.COMPACT_BLOCK_FLOATS
    g = torch.Generator().manual_seed(5)
    for active in (0.1, 0.9):
        D.reset_allreduce_state()                                     # (a packing set agreed on for other data would be outgrown)
        flat = torch.zeros(2000 * B + 13)
        blocks = torch.rand(2000, generator=g) < active
        flat[:2000 * B] = (torch.randn(2000, B, generator=g) * blocks[:, None]).reshape(-1)
        flat[2000 * B:] = 3.0
        flat = flat.to(dev)
        ref = flat.clone()
        for mode in ("always", "never", "auto", "auto"):              # (the second "auto" packs with the capacity of the first)
            f = flat.clone()
            st = {}
            D._allreduce_flat(f, None, mode, st)                     # what allreduce_gradients runs when world > 1
            torch.cuda.synchronize()
            assert torch.equal(f, ref), (active, mode, st)
            assert st["mode"] == {"always": "compact", "never": "dense"}.get(mode, "compact" if active < 0.5 else "dense"), (mode, st)
    # ONE collective, no host wait: the packing set is known beforehand (GradientSupport); the check travels in the packed
    # buffer and is looked at by the next call / verify_pending
    D.reset_allreduce_state()
    mask = torch.zeros(2000, dtype=torch.uint8)
    mask[100:400] = 1
    sup = D.GradientSupport(mask.to(dev), 2000 * B + 13)
    assert sup.count == 300
    flat = torch.zeros(2000 * B + 13)
    flat[150 * B:350 * B] = torch.randn(200 * B, generator=g)
    flat[2000 * B:] = 2.0
    flat = flat.to(dev)
    ref = flat.clone()
    st = {}
    D._allreduce_flat(flat, None, "auto", st, sup)
    assert st["mode"] == "compact" and st["collectives"] == 1 and st["sent_floats"] == 300 * B + 13 + 1, st
    D.verify_pending()
    assert torch.equal(flat, ref)
    flat[1000 * B + 5] = 1.0                                          # outside the support: found, raised by the next look
    D._allreduce_flat(flat, None, "auto", {}, sup)
    try:
        D.verify_pending()
        raise AssertionError("a violated GradientSupport went unnoticed")
    except RuntimeError as e:
        assert "GradientSupport" in str(e)
    st = {}
    D._allreduce_flat(flat, None, "auto", st, sup, True)              # strict: looked at right away, dense sums
    assert st["mode"] == "dense" and st.get("outgrown"), st
    D.reset_allreduce_state()
    loss = torch.tensor(1.25, device=dev)
    dist.all_reduce(loss)
    assert float(loss) == 1.25
    m = torch.tensor([0, 1, 0, 7], dtype=torch.uint8, device=dev)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    assert m.tolist() == [0, 1, 0, 7]
    u.allreduce_gradients({"_flat": torch.ones(128, device=dev)})   # world 1: returns without communicating
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_SMOKE_OK")


if __name__ == "__main__":
    main()
