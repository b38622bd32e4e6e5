"""What the compacted gradient all-reduce (distributed._allreduce_flat) would move, and what its packing costs,
measured on ONE GPU: the gradient of one H1 step of the headline scene is split as `world` ranks would compute it
(interleaved pixel chunks), the union of the non-zero 1-KiB blocks is formed exactly as the collective would, and the
mask / pack / unpack kernels are timed.  The collective itself needs `world` GPUs; DESIGN.md section 7 prices it.

    python tools/allreduce_compaction.py [--world 8] [--majorant-factor 0|8]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--majorant-factor", type=int, default=0)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--film", type=int, default=512)
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--block", type=int, default=0, help="block size in floats (0 = the product's)")
    args = ap.parse_args()
    import torch
    import uivr_amd as u
    from uivr_amd import synthetic, distributed as D
    dev = torch.device("cuda", 0)
    scene = synthetic.dust_devil_scene(res=args.res, film=args.film, device=dev)
    scene.medium.majorant_resolution_factor = args.majorant_factor
    sensor = scene.sensors[0]
    n_pixels = sensor.width * sensor.height
    integ = u.get_int_config("volpathsimple-drt").create(max_depth=64)
    B = args.block or D.COMPACT_BLOCK_FLOATS
    union = None
    fr = []
    for r in range(args.world):
        shard = u.ShardSpec(r, args.world, u.ShardSpec.default_chunk(n_pixels, args.world))
        off, inter = shard.ray_mapping(args.spp)
        batch = u.RayBatch(n_rays=shard.n_local_pixels(n_pixels) * args.spp, spp=args.spp, sensor=sensor, ray_offset=off, interleave=inter)
        grads = u.alloc_grads(scene)
        sampler = u.IndependentSampler(u.sample_tea_32(1, 988378)[0], args.spp)
        L, _, state = integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)
        img = integ.develop(scene, L, args.spp)
        dL = integ.film_backward(scene, (2.0 / (n_pixels * 3)) * (img - 0.5), args.spp)
        integ.sample(u.ADMode.Backward, scene, sampler, batch, δL=dL, state_in=state, grads=grads)
        flat = grads["_flat"]
        n_full = flat.numel() // B * B
        m = (flat[:n_full].view(-1, B) != 0).any(1)
        fr.append(float(m.float().mean()))
        union = m if union is None else (union | m)
    V = args.res ** 3
    vb = V // B
    out = {"world": args.world, "majorant_resolution_factor": args.majorant_factor, "block_floats": B,
           "buffer_MiB": flat.numel() * 4 / 2 ** 20,
           "active_fraction_per_rank": [round(x, 4) for x in fr],
           "active_fraction_union": round(float(union.float().mean()), 4),
           "active_fraction_union_sigma_t_plane": round(float(union[:vb].float().mean()), 4),
           "active_fraction_union_albedo_planes": round(float(union[vb:].float().mean()), 4)}
    # cost of the packing around the collective (last rank's buffer; the same kernels the product path runs)
    body = flat[:n_full].view(-1, B)

    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    out["t_mask_ms"] = round(timed(lambda: D._block_mask(body)), 4)
    out["t_mask_any_ms"] = round(timed(lambda: (body != 0).any(dim=1).to(torch.uint8)), 4)
    assert torch.equal(D._block_mask(body), (body != 0).any(dim=1).to(torch.uint8))
    mask = union.to(torch.uint8)
    out["t_nonzero_with_host_sync_ms"] = round(timed(lambda: mask.nonzero(as_tuple=False).squeeze(1).numel()), 4)
    idx = mask.nonzero(as_tuple=False).squeeze(1)
    out["t_pack_ms"] = round(timed(lambda: body.index_select(0, idx)), 4)
    packed = body.index_select(0, idx)
    out["t_unpack_ms"] = round(timed(lambda: body.index_copy_(0, idx, packed)), 4)
    out["packed_MiB"] = round(packed.numel() * 4 / 2 ** 20, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
