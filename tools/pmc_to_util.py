#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (tools/pmc_util.txt pass) + the kernel-trace summary -> "_util:<key>" of profiles/roofline_traffic.json:
the SECONDARY ceilings SURVEY.md 8d asks for next to the HBM roofline, per kernel of the step -

  lanes_active = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)      (share of the 64 lanes a vector instruction has work for)
  valu_busy    = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles)   (SQ_ACTIVE_INST_* count quad-cycles, MI355X_MICROARCH.md;
                 kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs, or SQ_BUSY_CYCLES / 32 shader engines where that counter is missing)
  avg_ms       = the kernel's average duration in the rocprofv3 --kernel-trace --stats pass of the same command
  ms_per_step  = avg_ms x (the kernel's launches per launch of a primal tracer kernel, i.e. per step of the command)

    python tools/pmc_to_util.py <dir with *counter_collection.csv | summary .txt of tools/pmc_summary.py> <kernel_stats.csv> <workload key> [out.json]

Hash-gated like the traffic figures (bench.kernel_source_sha16): bench.py quotes them only for the sources they came from.
"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16

root, stats, key = sys.argv[1], sys.argv[2], sys.argv[3]
out = sys.argv[4] if len(sys.argv) > 4 else "profiles/roofline_traffic.json"


def short(k):
    return k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


acc = collections.defaultdict(lambda: collections.defaultdict(list))
if os.path.isfile(root):                               # a summary written by tools/pmc_summary.py: "kernel" lines, "   COUNTER  mean  (n=..)" lines
    cur = None
    for line in open(root):
        if not line.startswith(" "):
            cur = line.strip()
        elif cur and line.split():
            parts = line.split()
            acc[cur][parts[0]].append(float(parts[1]))
for f in ([] if os.path.isfile(root) else glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "drt::" in r["Kernel_Name"]:
            per[(short(r["Kernel_Name"]), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
def is_sq_tail(k):
    """trace_sq_kernel<ADJ, COUNT, ENV, MG, QUAD, TAILM[, ROUNDS]>: the tail launch of a step (TAILM) is a second launch, not a step."""
    if "trace_sq_kernel<" not in k:
        return False
    args = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
    return len(args) > 5 and args[5] == "true"


ms, calls = {}, {}
for r in csv.DictReader(open(stats)):
    ms[short(r["Name"])] = float(r["AverageNs"]) / 1e6
    calls[short(r["Name"])] = int(r["Calls"])
# launches of a primal tracer kernel (counting instantiation included) = steps of the profiled command
# (the queued tracer's tail launches - last template argument true - are second launches of a step, not steps)
steps = sum(c for k, c in calls.items() if any(t in k for t in ("trace_sq_kernel<false", "trace_coop_kernel<false", "trace_super_kernel<false",
                                                                "trace_wavefront_kernel<false"))
            and not is_sq_tail(k))
util = {}
for k in acc:
    m = lambda c: (sum(acc[k][c]) / len(acc[k][c])) if acc[k].get(c) else 0.0
    a, t, b, g = m("SQ_ACTIVE_INST_VALU"), m("SQ_THREAD_CYCLES_VALU"), m("SQ_BUSY_CYCLES"), m("GRBM_GUI_ACTIVE")
    cycles = g / 8.0 if g > 0 else b / 32.0
    if a <= 0 or cycles <= 0:
        continue
    util[k] = {"lanes_active": round(t / (64.0 * a), 4), "valu_busy": round(4.0 * a / (1024.0 * cycles), 4),
               "valu_insts": m("SQ_INSTS_VALU"), "salu_insts": m("SQ_INSTS_SALU"), "avg_ms": round(ms.get(k, 0.0), 4) or None,
               "ms_per_step": (round(ms[k] * calls[k] / steps, 4) if k in ms and steps else None)}
sha = kernel_source_sha16()
res = {}
if os.path.exists(out):
    try:
        old = json.load(open(out))
        if old.get("source_sha16") == sha:
            res = old
    except Exception:
        res = {}
res["source_sha16"] = sha
res["_util:" + key] = util
res["_util_method"] = __doc__.split("\n\n")[1]
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(util.items(), key=lambda kv: -(kv[1]["avg_ms"] or 0)):
    if (v["avg_ms"] or 0) > 0.05:
        print(f"{k:78s} {v['avg_ms']:8.3f} ms  lanes active {v['lanes_active']:.3f}  VALU busy {v['valu_busy']:.3f}")
