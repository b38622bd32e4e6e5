#!/usr/bin/env python3
"""Where does the L2-miss traffic of the tracer launches go?  (round-4 review, item 2a)

Hardware counters do not attribute traffic to buffers.  What can be had:
  * the request-size counter totals per kernel (tools/pmc_traffic.txt passes; tools/pmc_to_traffic.py) of the production build and of ONE
    ablation build whose computation is identical: -DDRT_EXP_ALB=1 returns the albedo without loading it - the headline scene's albedo IS
    the constant it returns, so every path, walk, lookup and record is the same and the difference of the totals is the albedo's traffic;
  * exact byte counts of the streams that are written / read once: splat records (event counters x record sizes), L / dL / L_in, ray keys;
  * the path cache from the iteration counts (entries written by the primal pass, read by the adjoint pass);
  * the records' global halves (Params::sq_cold) from the WRITE side: the adjoint tracer writes nothing else besides its splat records, so
    writes - records = write-backs of evicted cold lines, and a line that was evicted dirty is read back the next time its ray has a
    transition (reads ~= writes);
  * the sigma_t bricks: what is left of the reads.

    python tools/traffic_by_buffer.py <traffic.json of the production build> <traffic.json of the alb0 build> <bench.json> <workload key>
"""
import json, sys

base, alb0, bench, key = (json.load(open(p)) if i < 3 else p for i, p in enumerate(sys.argv[1:5]))
det_b, det_a = base["_detail:" + key], alb0["_detail:" + key]


def find(det, frag):
    # (the adjoint has two instantiations per launch since round 5: the main launch and the tail launch that finishes the tail pool's records)
    ks = [k for k in det if frag in k]
    assert 1 <= len(ks) <= 2, (frag, ks)
    return {f: sum(det[k][f] for k in ks) for f in ("read_bytes", "write_bytes", "total")}


n = bench["config"]["n_samples_per_step"] if "n_samples_per_step" in bench.get("config", {}) else 512 * 512 * 32
cp, ca = bench["counters_primal"], bench["counters_adjoint"]
GB = 1e9
rows = []
for tag, frag, cnt in (("primal tracer", "trace_sq_kernel<false, false", cp), ("adjoint tracer", "trace_sq_kernel<true, false", ca)):
    b, a = find(det_b, frag), find(det_a, frag)
    rd, wr = b["read_bytes"], b["write_bytes"]
    alb_rd = max(0.0, rd - a["read_bytes"])
    out = {"kernel": tag, "read_GB": rd / GB, "write_GB": wr / GB, "albedo_read_GB": alb_rd / GB,
           "albedo_lookups_M": cnt["n_alb"] / 1e6, "albedo_bytes_per_lookup": alb_rd / max(1, cnt["n_alb"])}
    if tag.startswith("primal"):
        io_w = 12 * n + 4 * n + 1 * n                              # L_out, ray_hash, ray_iters
        out["L_out_hash_iters_write_GB"] = io_w / GB
        out["path_cache_and_cold_write_GB"] = (wr - io_w) / GB     # path-cache entries (16 B each, one or two per bounce-loop iteration) + evicted cold lines
        out["sigma_t_and_cold_read_GB"] = (rd - alb_rd) / GB
    else:
        rec = 16 * (cnt["n_tr"] + cnt["n_rt_adj"] + cnt["n_sc"] - cnt["n_sc_alb"]) + 32 * cnt["n_sc_alb"]   # (zero-valued splats are not emitted: an upper bound)
        io_r = 24 * n + 4 * n + 1 * n                              # dL, L_in, ray_hash, ray_iters (the order's keys)
        cold_w = max(0.0, wr - rec)
        out.update({"records_write_GB": rec / GB, "cold_writeback_GB": cold_w / GB, "cold_reread_GB_est": cold_w / GB,
                    "dL_Lin_keys_read_GB": io_r / GB,
                    "sigma_t_bricks_and_path_cache_read_GB": (rd - alb_rd - cold_w - io_r) / GB,
                    "sigma_t_lookups_M": (cnt["n_dt"] + cnt["n_rt"] + cnt["n_drt"]) / 1e6})
    rows.append(out)
print(f"L2-miss traffic by buffer, workload {key} (GB per launch; request-size counters, see tools/pmc_to_traffic.py)\n")
for r in rows:
    print(r["kernel"])
    for k, v in r.items():
        if k != "kernel":
            print(f"    {k:42s} {v:10.3f}")
    print()
