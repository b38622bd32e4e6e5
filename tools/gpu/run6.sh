cd /root/repo
mkdir -p gpurun_out/r4f
DRT_PROFILE_MODE=sq LD_LIBRARY_PATH=variants/sqprof python tools/super_profile.py 2>&1 | tail -2 | tee gpurun_out/r4f/sqprof.txt
DRT_PROFILE_MODE=sq2 LD_LIBRARY_PATH=variants/sqprof2 python tools/super_profile.py 2>&1 | tail -2 | tee gpurun_out/r4f/sqprof2.txt
bash tools/gpu/pmc8.sh r4f_pmc8 2>&1 | tail -60
