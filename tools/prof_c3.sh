R=/root/repo/gpurun_out/c3prof
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof -o c3 -- python /root/repo/tools/bench_optimize.py --iters 60 --ref-spp 16 > $R/c3.json 2> $R/err.txt)
cd /root/repo
python tools/rocpd_stats.py $R/prof/c3_results.db --csv $R/c3_kernel_stats.csv --top 25
rm -rf $R/prof
cat $R/c3.json
