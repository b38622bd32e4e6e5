# round 5, call AA: SOLO tails (the tail launch runs its records to their ends in registers): parity, then timing
cd /root/repo; mkdir -p gpurun_out/r5aa
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_estimators.py tests/test_gpu_envmap.py -m gpu -x -q > gpurun_out/r5aa/pytest_a.txt 2>&1; tail -n 6 gpurun_out/r5aa/pytest_a.txt
bash tools/gpu/sweep2.sh default nosolo soloprimal default nosolo > gpurun_out/r5aa/sweep.txt 2>&1; cat gpurun_out/r5aa/sweep.txt
for v in nosolo soloprimal; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5aa/share_$v.txt 2>&1; done
timeout 200 python tools/gpu/share.py > gpurun_out/r5aa/share_default.txt 2>&1
tail -qn 1 gpurun_out/r5aa/share_*.txt
