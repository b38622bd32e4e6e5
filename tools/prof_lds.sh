#!/bin/bash
# Usage (GPU box, repo root): tools/prof_lds.sh <config|headline>  -> LDS / memory-wait counters per kernel (rocprofv3 --pmc with --kernel-trace only)
cfg=${1:-config5_nerf_256_512x32}
R=/root/repo/gpurun_out/lds_$cfg
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
if [ "$cfg" == "headline" ]; then A="--no-cpu-baseline --no-extra-configs --steps 3 --warmup 1"; else A="--only-config $cfg"; fi
(timeout 900 rocprofv3 -i /root/repo/tools/pmc_lds.txt --kernel-trace --output-format csv -d $R/pmc -- python /root/repo/bench.py $A > /dev/null 2> $R/err.txt)
cd /root/repo
python tools/pmc_summary.py $R/pmc > $R/pmc_lds.txt
rm -rf $R/pmc
grep -A17 "bin_histogram\|bin_scatter\|tile_reduce" $R/pmc_lds.txt | head -80
