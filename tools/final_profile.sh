#!/bin/bash
# Usage (GPU box, repo root): tools/final_profile.sh <tag>
#   -> gpurun_out/<tag>/{kernel_stats.csv,kernel_stats_factor8.csv,pmc_util.txt,factor8_pmc_util.txt,pmc_traffic.txt,
#                        pmc_traffic_factor8.txt,roofline_traffic.json,bench.json,...}
# The round's judged profile set: rocprofv3 kernel traces, utilisation and traffic counter passes (separate runs, --pmc with
# --kernel-trace only), for the headline (global majorant) and for the reference's default majorant_resolution_factor 8, then
# the full bench line (which quotes roofline_traffic.json only because it was taken from the same kernel sources).
tag=${1:-final}
R=/root/repo/gpurun_out/$tag
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs"
for f in 0 8; do
  sfx=""; [ $f == 8 ] && sfx="_factor8"
  (timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof$f -o hl -- $B --majorant-factor $f > $R/bench_under_rocprof$sfx.json 2>> $R/err.txt)
  (timeout 600 rocprofv3 -i /root/repo/tools/pmc_util.txt --kernel-trace --output-format csv -d $R/pmc_util$f -- $B --steps 3 --warmup 1 --majorant-factor $f > /dev/null 2>> $R/err.txt)
  (timeout 900 rocprofv3 -i /root/repo/tools/pmc_traffic.txt --kernel-trace --output-format csv -d $R/pmc_traffic$f -- $B --steps 3 --warmup 1 --majorant-factor $f > /dev/null 2>> $R/err.txt)
done
(timeout 900 rocprofv3 -i /root/repo/tools/pmc_traffic.txt --kernel-trace --output-format csv -d $R/pmc_traffic_fused -- $B --only-config config5_fused_nerf_drt_256_512x32 > /dev/null 2>> $R/err.txt)
cd /root/repo
rm -f $R/roofline_traffic.json
python tools/rocpd_stats.py $R/prof0/hl_results.db --csv $R/kernel_stats.csv --top 14
python tools/rocpd_stats.py $R/prof8/hl_results.db --csv $R/kernel_stats_factor8.csv --top 14
python tools/pmc_summary.py $R/pmc_util0 > $R/pmc_util.txt
python tools/pmc_summary.py $R/pmc_util8 > $R/factor8_pmc_util.txt
python tools/pmc_to_traffic.py $R/pmc_traffic0 dust-devil-256-512x32 $R/roofline_traffic.json > $R/pmc_traffic.txt
python tools/pmc_to_traffic.py $R/pmc_traffic8 dust-devil-256-512x32-factor8 $R/roofline_traffic.json > $R/pmc_traffic_factor8.txt
python tools/pmc_to_traffic.py $R/pmc_traffic_fused fused-256-512x32 $R/roofline_traffic.json > $R/pmc_traffic_fused.txt
rm -rf $R/pmc_util0 $R/pmc_util8 $R/pmc_traffic0 $R/pmc_traffic8 $R/pmc_traffic_fused $R/prof0 $R/prof8
cp $R/roofline_traffic.json profiles/roofline_traffic.json      # the bench lines below quote it (same kernel sources: hash checked)
(timeout 1500 python bench.py > $R/bench.json 2>> $R/err.txt)
(timeout 600 python bench.py --majorant-factor 8 --no-extra-configs --no-cpu-baseline > $R/bench_factor8.json 2>> $R/err.txt)
python - <<P
import json
for f in ("bench.json", "bench_factor8.json"):
    d=json.load(open("$R/" + f)); print(f, d["value"], d["t_primal_ms"], d["t_adjoint_ms"], d["t_grad_reduce_ms"], d["roofline"]["frac"], d["roofline"]["traffic"])
d=json.load(open("$R/bench.json"))
print({k:(v.get("value"),v.get("error")) for k,v in d["other_configs"].items()})
P
