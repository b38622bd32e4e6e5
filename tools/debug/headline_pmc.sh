# counters of the headline's kernels: bash tools/debug/headline_pmc.sh COUNTERFILE(tools/NAME.txt) [VARIANT] -> gpurun_out/hlpmc/NAME_VARIANT.txt
cd /tmp && export TMPDIR=/tmp
f=$1; v=${2:-default}; L=""; [ "$v" != "default" ] && L=/root/repo/variants/$v
mkdir -p /root/repo/gpurun_out/hlpmc
o=/root/repo/gpurun_out/hlpmc/${f}_$v
LD_LIBRARY_PATH=$L timeout 600 rocprofv3 -i /root/repo/tools/$f.txt --kernel-trace --output-format csv -d $o -- python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1 > /dev/null 2>&1
python /root/repo/tools/pmc_summary.py $o > $o.txt 2>&1
rm -rf $o
for k in "trace_sq_kernel<true, false, false, false, false, false, false>" "trace_sq_kernel<false, false, false, false, false, false, false>" tile_reduce_kernel; do grep -F -A17 "$k" $o.txt | head -18; done
