#!/bin/bash
# Usage (GPU box, repo root): tools/prof_small.sh [spp]   -> kernel trace of the headline scene at a small launch size (an eighth of the headline at spp 4)
spp=${1:-4}
R=/root/repo/gpurun_out/small
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof -o s -- python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --spp $spp --steps 10 --warmup 3 > $R/bench.json 2> $R/err.txt)
cd /root/repo
python tools/rocpd_stats.py $R/prof/s_results.db --csv $R/kernel_stats.csv --top 30
rm -rf $R/prof
