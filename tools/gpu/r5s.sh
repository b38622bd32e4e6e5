# counters of the nerf tile adjoint on config 5 (nerf alone)
cd /root/repo; mkdir -p gpurun_out/r5s
R=/root/repo/gpurun_out/r5s
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --only-config config5_nerf_256_512x32"
(timeout 600 rocprofv3 -i /root/repo/tools/pmc_lds.txt --kernel-trace --output-format csv -d $R/l -- $B > /dev/null 2> $R/err.txt)
(timeout 600 rocprofv3 -i /root/repo/tools/pmc_util.txt --kernel-trace --output-format csv -d $R/u -- $B > /dev/null 2>> $R/err.txt)
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/s -- $B > /dev/null 2>> $R/err.txt)
cd /root/repo
python tools/pmc_summary.py $R/l > $R/pmc_lds.txt 2>&1
python tools/pmc_summary.py $R/u > $R/pmc_util.txt 2>&1
cat $R/s/*/*kernel_stats.csv | head -8 | cut -c1-200 > $R/kernel_stats.txt
rm -rf $R/l $R/u $R/s
grep -A18 "nerf_tile" $R/pmc_lds.txt | head -40
grep -A20 "nerf_tile" $R/pmc_util.txt | head -40
cat $R/kernel_stats.txt
