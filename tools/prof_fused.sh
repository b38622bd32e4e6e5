#!/bin/bash
# Usage (GPU box, repo root): tools/prof_fused.sh [config]  -> kernel trace of one other_configs entry (default: the fused nerf + DRT pass)
cfg=${1:-config5_fused_nerf_drt_256_512x32}
R=/root/repo/gpurun_out/fused
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof -o s -- python /root/repo/bench.py --only-config $cfg > $R/bench.json 2> $R/err.txt)
cd /root/repo
python tools/rocpd_stats.py $R/prof/s_results.db --csv $R/kernel_stats.csv --top 12
rm -rf $R/prof
