#!/bin/bash
# Usage (GPU box, repo root): tools/final_profile.sh <tag>   -> gpurun_out/<tag>/{kernel_stats.csv,pmc_util.txt,pmc_traffic.txt,roofline_traffic.json,bench.json,...}
# The round's judged profile set: rocprofv3 kernel trace, utilisation and traffic counter passes (separate runs), full bench line.
tag=${1:-final}
R=/root/repo/gpurun_out/$tag
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs"
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof -o hl -- $B > $R/bench_under_rocprof.json 2> $R/err.txt)
(timeout 600 rocprofv3 -i /root/repo/tools/pmc_util.txt --kernel-trace --output-format csv -d $R/pmc_util -- $B --steps 3 --warmup 1 > /dev/null 2>> $R/err.txt)
(timeout 900 rocprofv3 -i /root/repo/tools/pmc_traffic.txt --kernel-trace --output-format csv -d $R/pmc_traffic -- $B --steps 3 --warmup 1 > /dev/null 2>> $R/err.txt)
(timeout 600 rocprofv3 -i /root/repo/tools/pmc_util.txt --kernel-trace --output-format csv -d $R/pmc_util8 -- $B --steps 3 --warmup 1 --majorant-factor 8 > /dev/null 2>> $R/err.txt)
cd /root/repo
python tools/rocpd_stats.py $R/prof/hl_results.db --csv $R/kernel_stats.csv --top 12
python tools/pmc_summary.py $R/pmc_util > $R/pmc_util.txt
python tools/pmc_summary.py $R/pmc_util8 > $R/factor8_pmc_util.txt
python tools/pmc_to_traffic.py $R/pmc_traffic dust-devil-256-512x32 $R/roofline_traffic.json > $R/pmc_traffic.txt
rm -rf $R/pmc_util $R/pmc_util8 $R/pmc_traffic $R/prof
cp $R/roofline_traffic.json profiles/roofline_traffic.json      # the bench line below reads it
(timeout 1200 python bench.py > $R/bench.json 2>> $R/err.txt)
python - <<P
import json
d=json.load(open("$R/bench.json")); print(d["value"], d["t_primal_ms"], d["t_adjoint_ms"], d["t_grad_reduce_ms"], d["roofline"]["frac"], d["roofline"]["traffic"])
print({k:(v.get("value"),v.get("error")) for k,v in d["other_configs"].items()})
P
