#!/bin/bash
# Usage: tools/save_profiles.sh <gpurun_out tag> <round prefix, e.g. r05>   -> the judged copies under profiles/ (the scratch set stays in gpurun_out/)
t=gpurun_out/$1; p=profiles/$2
cp $t/bench.json ${p}_final_bench.json; cp $t/bench_under_rocprof.json ${p}_final_bench_under_rocprof.json
cp $t/kernel_stats.csv ${p}_final_kernel_stats.csv; cp $t/pmc_traffic.txt ${p}_final_pmc_traffic.txt; cp $t/pmc_util.txt ${p}_final_pmc_util.txt
cp $t/bench_factor0.json ${p}_factor0_bench.json; cp $t/bench_under_rocprof_factor0.json ${p}_factor0_bench_under_rocprof.json
cp $t/kernel_stats_factor0.csv ${p}_factor0_kernel_stats.csv; cp $t/pmc_traffic_factor0.txt ${p}_factor0_pmc_traffic.txt; cp $t/factor0_pmc_util.txt ${p}_factor0_pmc_util.txt
cp $t/pmc_traffic_envmap8.txt ${p}_envmap8_pmc_traffic.txt
[ -f $t/pmc_traffic_config4.txt ] && cp $t/pmc_traffic_config4.txt ${p}_config4_pmc_traffic.txt
cp $t/kernel_stats_fused.csv ${p}_fused_kernel_stats.csv; cp $t/pmc_traffic_fused.txt ${p}_fused_pmc_traffic.txt; cp $t/fused_pmc_util.txt ${p}_fused_pmc_util.txt
cat $t/util.txt $t/util_factor0.txt $t/util_fused.txt > ${p}_util_summary.txt
[ -f $t/traffic_by_buffer.txt ] && cp $t/traffic_by_buffer.txt ${p}_traffic_by_buffer_counters.txt
cp $t/roofline_traffic.json profiles/roofline_traffic.json
python - <<P
import json
d = json.load(open("$t/bench.json"))
json.dump(d["other_configs"].get("headline_rank_share"), open("${p}_rank_share.json", "w"), indent=1)
json.dump({k: d["other_configs"].get(k) for k in ("config3_optimize_loop", "config3_as_reproduce")}, open("${p}_config3.json", "w"), indent=1)
P
ls ${p}_*
