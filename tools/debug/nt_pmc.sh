cd /tmp && export TMPDIR=/tmp
for v in ntspw1 ntspw4; do
  for f in pmc_lds pmc_util; do
    LD_LIBRARY_PATH=/root/repo/variants/$v timeout 600 rocprofv3 -i /root/repo/tools/$f.txt --kernel-trace --output-format csv -d /root/repo/gpurun_out/ntpmc/${v}_$f -- python /root/repo/bench.py --only-config config5_nerf_256_512x32 > /dev/null 2>&1
    python /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/ntpmc/${v}_$f > /root/repo/gpurun_out/ntpmc/${v}_$f.txt 2>&1
    rm -rf /root/repo/gpurun_out/ntpmc/${v}_$f
  done
  echo "=== $v"; grep -A22 "nerf_tile_adjoint_kernel<true>" /root/repo/gpurun_out/ntpmc/${v}_pmc_lds.txt | head -24; grep -A20 "nerf_tile_adjoint_kernel<true>" /root/repo/gpurun_out/ntpmc/${v}_pmc_util.txt | head -22
done
