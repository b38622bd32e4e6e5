#!/bin/bash
# Usage (GPU box, repo root): tools/prof8.sh <tag> [extra bench args]  -> gpurun_out/<tag>/{bench8.json,kernel_stats8.csv,pmc_util8.txt}
# Headline scene at the reference's majorant_resolution_factor 8: bench line, rocprofv3 kernel trace, utilisation counters.
tag=${1:-p8}; shift
R=/root/repo/gpurun_out/$tag
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --majorant-factor 8 $@"
(timeout 300 $B --steps 5 --warmup 2 > $R/bench8.json 2> $R/err.txt)
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof -o hl -- $B --steps 5 --warmup 2 > /dev/null 2>> $R/err.txt)
(timeout 600 rocprofv3 -i /root/repo/tools/pmc_util.txt --kernel-trace --output-format csv -d $R/pmc_util8 -- $B --steps 3 --warmup 1 > /dev/null 2>> $R/err.txt)
cd /root/repo
python tools/rocpd_stats.py $R/prof/hl_results.db --csv $R/kernel_stats8.csv --top 12 > /dev/null
python tools/pmc_summary.py $R/pmc_util8 > $R/pmc_util8.txt
rm -rf $R/pmc_util8 $R/prof
python - <<P
import json
d=json.load(open("$R/bench8.json")); print(d["value"], d["t_primal_ms"], d["t_adjoint_ms"], d["t_grad_reduce_ms"])
P
