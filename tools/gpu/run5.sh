cd /root/repo
mkdir -p gpurun_out/r4e
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "supergrid or majorant" > gpurun_out/r4e/t1.txt 2>&1; echo "rc $?" >> gpurun_out/r4e/t1.txt
tail -3 gpurun_out/r4e/t1.txt
bash tools/sweep_variants.sh "--majorant-factor 8 --steps 10 --warmup 3" default
DRT_PROFILE_MODE=sq LD_LIBRARY_PATH=variants/sqprof python tools/super_profile.py 2>&1 | tail -2;DRT_PROFILE_MODE=sq2 LD_LIBRARY_PATH=variants/sqprof2 python tools/super_profile.py 2>&1 | tail -2
