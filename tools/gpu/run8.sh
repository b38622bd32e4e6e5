cd /root/repo
mkdir -p gpurun_out/r4h
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envmap.py -x -q -k "supergrid or majorant or envmap" > gpurun_out/r4h/t1.txt 2>&1; echo "rc $?" >> gpurun_out/r4h/t1.txt
tail -3 gpurun_out/r4h/t1.txt
bash tools/gpu/sweep2.sh default 2>&1 | tee gpurun_out/r4h/sweep.txt
for m in 2 3 4; do DRT_PROFILE_MODE=sq$m LD_LIBRARY_PATH=variants/sqprof$m python tools/super_profile.py 2>&1 | tail -2 | tee -a gpurun_out/r4h/sqprof.txt; done
