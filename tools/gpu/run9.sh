cd /root/repo
mkdir -p gpurun_out/r4i
DRT_PROFILE_MODE=sq LD_LIBRARY_PATH=variants/sqprof python tools/super_profile.py 2>&1 | tail -2 | tee gpurun_out/r4i/sqprof.txt
bash tools/gpu/pmc8.sh r4i_pmc8 > /dev/null 2>&1
grep -A19 "trace_sq_kernel<true, false, false>\|trace_sq_kernel<false, false, false>" gpurun_out/r4i_pmc8/pmc_util.txt | grep -v "^--"
grep -A17 "trace_sq_kernel<true, false, false>" gpurun_out/r4i_pmc8/pmc_lds.txt | grep -v "^--"
