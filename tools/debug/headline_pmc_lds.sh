# LDS counters of the headline's kernels (tile_reduce above all): bash tools/debug/headline_pmc_lds.sh [VARIANT] -> gpurun_out/hlpmc/
cd /tmp && export TMPDIR=/tmp
v=${1:-default}; L=""; [ "$v" != "default" ] && L=/root/repo/variants/$v
mkdir -p /root/repo/gpurun_out/hlpmc
LD_LIBRARY_PATH=$L timeout 600 rocprofv3 -i /root/repo/tools/pmc_lds.txt --kernel-trace --output-format csv -d /root/repo/gpurun_out/hlpmc/$v -- python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1 > /dev/null 2>&1
python /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/hlpmc/$v > /root/repo/gpurun_out/hlpmc/$v.txt 2>&1
rm -rf /root/repo/gpurun_out/hlpmc/$v
for k in tile_reduce_kernel bin_scatter_kernel bin_histogram_kernel "trace_sq_kernel<true, false, false, false, false, false, false>" "trace_sq_kernel<false, false, false, false, false, false, false>"; do grep -F -A17 "$k" /root/repo/gpurun_out/hlpmc/$v.txt | head -18; done
