#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (tools/pmc_traffic.txt passes) -> profiles/roofline_traffic.json.

HBM-side bytes per launch = 128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B + 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B)
(request-size counters; FETCH_SIZE under-reports 128-B requests by 2x on gfx950, MI355X_MICROARCH.md).
The adjoint PASS is the adjoint tracer plus the gradient reduction kernels that finish its splats.

    python tools/pmc_to_traffic.py <dir with *counter_collection.csv> <workload key> [out.json]

An existing out.json taken from the SAME kernel sources (source_sha16, bench.kernel_source_sha16) is extended with the new
key; one from other sources is replaced - bench.py quotes the file only when the hash matches the sources it runs.
"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16

root, key = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else "profiles/roofline_traffic.json"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "drt::" not in k:
            continue
        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        per[(short, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)


def mean(k, c):
    v = acc[k].get(c, [])
    return sum(v) / len(v) if v else 0.0


detail = {}
for k in acc:
    rd = 128 * mean(k, "TCC_EA0_RDREQ_128B_sum") + 64 * mean(k, "TCC_EA0_RDREQ_64B_sum") + 32 * mean(k, "TCC_EA0_RDREQ_32B_sum")
    wr = 64 * mean(k, "TCC_EA0_WRREQ_64B_sum") + 32 * (mean(k, "TCC_EA0_WRREQ_sum") - mean(k, "TCC_EA0_WRREQ_64B_sum"))
    detail[k] = {"read_bytes": rd, "write_bytes": wr, "total": rd + wr, "fetch_size_kib": mean(k, "FETCH_SIZE"),
                 "write_size_kib": mean(k, "WRITE_SIZE"), "atomics": mean(k, "TCC_EA0_ATOMIC_sum"),
                 "launches_seen": len(acc[k].get("TCC_EA0_RDREQ_sum", []))}
adj = [k for k in detail if "trace_kernel<true, false" in k or "trace_coop_kernel<true, false" in k or "trace_super_kernel<true, false" in k or "trace_sq_kernel<true, false" in k or "nerf_tile_" in k or
       "bin_" in k or "tile_reduce" in k or "untile" in k]      # (nerf_tile_*: the nerf half of the fused pass - adjoint kernel + its bounds reduction)
pri = [k for k in detail if "trace_wavefront_kernel<false, false" in k or "trace_coop_kernel<false, false" in k or "trace_super_kernel<false, false" in k or
       "trace_sq_kernel<false, false" in k or "nerf_kernel<false, false" in k]
if key.startswith("fused"):
    # (bench.py's fused entry also runs its envmap + factor-8 variant in the same command: the key is the plain variant - the nerf kernels, the
    #  wave-cooperative tracer, the reduction - without the queued tracer's launches of the variant)
    adj = [k for k in adj if "trace_sq_kernel" not in k]
    pri = [k for k in pri if "trace_sq_kernel" not in k]
if key.startswith("config4"):
    # (bench.py's config-4 entry runs the global-majorant variant in the same command: the key is the factor-8 run - the queued tracer's
    #  kernels; the reduction kernels are averaged over both variants' launches, which move the same records)
    adj = [k for k in adj if "trace_coop_kernel" not in k]
    pri = [k for k in pri if "trace_coop_kernel" not in k]
sha = kernel_source_sha16()
res = {}
if os.path.exists(out):
    try:
        old = json.load(open(out))
        if old.get("source_sha16") == sha:
            res = old
    except Exception:
        res = {}
res.update({"source_sha16": sha, key: sum(detail[k]["total"] for k in adj),
            key + ":primal": sum(detail[k]["total"] for k in pri),
            "_adjoint_pass_kernels:" + key: adj, "_detail:" + key: detail,
            "_method": __doc__.split("\n\n")[1]})
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))
for k in adj + pri:
    print(f"{k:70s} read {detail[k]['read_bytes'] / 1e9:7.2f} GB  write {detail[k]['write_bytes'] / 1e9:7.2f} GB  atomics {detail[k]['atomics']:.3g}")
