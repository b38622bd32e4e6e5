# round 5, call F: wide albedo loads, near-majorant order key, cache depth; config3_as_reproduce first run
cd /root/repo
mkdir -p gpurun_out/r5f
bash tools/gpu/sweep2.sh default rgbn near1 near2 near3 cap64 cap64n2 noue default rgbn > gpurun_out/r5f/sweep.txt 2>&1
cat gpurun_out/r5f/sweep.txt
for v in prof6 prof6n2; do LD_LIBRARY_PATH=variants/$v DRT_PROFILE_SPP=32 timeout 300 python tools/finish_age_profile.py > gpurun_out/r5f/finish_age_$v.txt 2>&1; done
timeout 900 python bench.py --only-config config3_as_reproduce > gpurun_out/r5f/c3r.json 2> gpurun_out/r5f/c3r.err; tail -c 1500 gpurun_out/r5f/c3r.json; tail -n 5 gpurun_out/r5f/c3r.err
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py -m gpu -x -q 2>&1 | tail -3
