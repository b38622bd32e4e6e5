cd /root/repo
mkdir -p gpurun_out/r4g
for m in 3 4; do DRT_PROFILE_MODE=sq$m LD_LIBRARY_PATH=variants/sqprof$m python tools/super_profile.py 2>&1 | tail -2 | tee -a gpurun_out/r4g/sqprof34.txt; done
bash tools/gpu/sweep2.sh default fm_div fm_divfma fm_log fm_all 2>&1 | tee gpurun_out/r4g/fastmath.txt
