#!/bin/bash
# Usage (GPU box, repo root): tools/final_profile.sh <tag>
#   -> gpurun_out/<tag>/{kernel_stats.csv,kernel_stats_factor0.csv,pmc_util.txt,factor0_pmc_util.txt,pmc_traffic.txt,
#                        pmc_traffic_factor0.txt,pmc_traffic_envmap8.txt,pmc_traffic_fused.txt,roofline_traffic.json,bench.json,...}
# The round's judged profile set: rocprofv3 kernel traces, utilisation and traffic counter passes (separate runs, --pmc with
# --kernel-trace only) for the headline at the reference's default majorant_resolution_factor 8 (the main line) and with the
# global majorant, traffic passes for the envmap + factor-8 set-up and the fused configuration, then the full bench line
# (which quotes roofline_traffic.json only because it was taken from the same kernel sources).
tag=${1:-final}
R=/root/repo/gpurun_out/$tag
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs"
for f in 8 0; do
  sfx=""; [ $f == 0 ] && sfx="_factor0"
  (timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof$f -o hl -- $B --majorant-factor $f > $R/bench_under_rocprof$sfx.json 2>> $R/err.txt)
  (timeout 600 rocprofv3 -i /root/repo/tools/pmc_util.txt --kernel-trace --output-format csv -d $R/pmc_util$f -- $B --steps 3 --warmup 1 --majorant-factor $f > /dev/null 2>> $R/err.txt)
  (timeout 900 rocprofv3 -i /root/repo/tools/pmc_traffic.txt --kernel-trace --output-format csv -d $R/pmc_traffic$f -- $B --steps 3 --warmup 1 --majorant-factor $f > /dev/null 2>> $R/err.txt)
done
# (traffic by buffer: the same pass over the build whose albedo lookups load nothing - tools/mk_variant.sh alb0 "-DDRT_EXP_ALB=1" drt_sq.hip, built before the call)
[ -f /root/repo/variants/alb0/libdrt_hip.so ] && (LD_LIBRARY_PATH=/root/repo/variants/alb0 timeout 900 rocprofv3 -i /root/repo/tools/pmc_traffic.txt --kernel-trace --output-format csv -d $R/pmc_traffic_alb0 -- $B --steps 3 --warmup 1 --majorant-factor 8 > /dev/null 2>> $R/err.txt)
(timeout 900 rocprofv3 -i /root/repo/tools/pmc_traffic.txt --kernel-trace --output-format csv -d $R/pmc_traffic_env8 -- python /root/repo/bench.py --only-config headline_envmap_factor8 > /dev/null 2>> $R/err.txt)
(timeout 900 rocprofv3 -i /root/repo/tools/pmc_traffic.txt --kernel-trace --output-format csv -d $R/pmc_traffic_fused -- python /root/repo/bench.py --only-config config5_fused_nerf_drt_256_512x32 > /dev/null 2>> $R/err.txt)
(timeout 900 rocprofv3 -i /root/repo/tools/pmc_traffic.txt --kernel-trace --output-format csv -d $R/pmc_traffic_c4 -- python /root/repo/bench.py --only-config config4_512_rank_share_1024x64 > /dev/null 2>> $R/err.txt)
(timeout 900 rocprofv3 -i /root/repo/tools/pmc_util.txt --kernel-trace --output-format csv -d $R/pmc_util_fused -- python /root/repo/bench.py --only-config config5_fused_nerf_drt_256_512x32 > /dev/null 2>> $R/err.txt)
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof_fused -o hl -- python /root/repo/bench.py --only-config config5_fused_nerf_drt_256_512x32 > /dev/null 2>> $R/err.txt)
cd /root/repo
rm -f $R/roofline_traffic.json
python tools/rocpd_stats.py $R/prof8/hl_results.db --csv $R/kernel_stats.csv --top 14
python tools/rocpd_stats.py $R/prof0/hl_results.db --csv $R/kernel_stats_factor0.csv --top 14
python tools/rocpd_stats.py $R/prof_fused/hl_results.db --csv $R/kernel_stats_fused.csv --top 14 > /dev/null
python tools/pmc_summary.py $R/pmc_util8 > $R/pmc_util.txt
python tools/pmc_summary.py $R/pmc_util0 > $R/factor0_pmc_util.txt
python tools/pmc_summary.py $R/pmc_util_fused > $R/fused_pmc_util.txt
python tools/pmc_to_traffic.py $R/pmc_traffic8 dust-devil-256-512x32-factor8 $R/roofline_traffic.json > $R/pmc_traffic.txt
python tools/pmc_to_traffic.py $R/pmc_traffic0 dust-devil-256-512x32 $R/roofline_traffic.json > $R/pmc_traffic_factor0.txt
python tools/pmc_to_traffic.py $R/pmc_traffic_env8 dust-devil-256-512x32-factor8-envmap2048 $R/roofline_traffic.json > $R/pmc_traffic_envmap8.txt
python tools/pmc_to_traffic.py $R/pmc_traffic_fused fused-256-512x32 $R/roofline_traffic.json > $R/pmc_traffic_fused.txt
python tools/pmc_to_traffic.py $R/pmc_traffic_c4 config4-512-1024x64-rank0of8-factor8 $R/roofline_traffic.json > $R/pmc_traffic_config4.txt
python tools/pmc_to_util.py $R/pmc_util.txt $R/kernel_stats.csv dust-devil-256-512x32-factor8 $R/roofline_traffic.json > $R/util.txt
python tools/pmc_to_util.py $R/factor0_pmc_util.txt $R/kernel_stats_factor0.csv dust-devil-256-512x32 $R/roofline_traffic.json > $R/util_factor0.txt
python tools/pmc_to_util.py $R/fused_pmc_util.txt $R/kernel_stats_fused.csv fused-256-512x32 $R/roofline_traffic.json > $R/util_fused.txt
[ -d $R/pmc_traffic_alb0 ] && python tools/pmc_to_traffic.py $R/pmc_traffic_alb0 dust-devil-256-512x32-factor8 $R/traffic_alb0.json > /dev/null
rm -rf $R/pmc_traffic_alb0
rm -rf $R/pmc_traffic_c4 $R/pmc_util0 $R/pmc_util8 $R/pmc_traffic0 $R/pmc_traffic8 $R/pmc_traffic_fused $R/pmc_traffic_env8 $R/pmc_util_fused $R/prof0 $R/prof8 $R/prof_fused
cp $R/roofline_traffic.json profiles/roofline_traffic.json      # the bench lines below quote it (same kernel sources: hash checked)
(timeout 1500 python bench.py > $R/bench.json 2>> $R/err.txt)
(timeout 600 python bench.py --majorant-factor 0 --no-extra-configs --no-cpu-baseline > $R/bench_factor0.json 2>> $R/err.txt)
[ -f $R/traffic_alb0.json ] && python tools/traffic_by_buffer.py $R/roofline_traffic.json $R/traffic_alb0.json $R/bench.json dust-devil-256-512x32-factor8 > $R/traffic_by_buffer.txt
python - <<P
import json
for f in ("bench.json", "bench_factor0.json"):
    d=json.load(open("$R/" + f)); print(f, d["value"], d["t_primal_ms"], d["t_adjoint_ms"], d["t_grad_reduce_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], (d["roofline"].get("secondary") or {}).get("tile_reduce"))
d=json.load(open("$R/bench.json"))
print({k:(v.get("value"),v.get("error")) for k,v in d["other_configs"].items()})
P
