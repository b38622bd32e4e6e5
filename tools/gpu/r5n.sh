# round 5, call N: tail pool v3 (flag = top bit of the dead count, hand-over behind the loop, tail launch = its own instantiation)
cd /root/repo
mkdir -p gpurun_out/r5n
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_estimators.py -m gpu -x -q > gpurun_out/r5n/pytest_a.txt 2>&1; tail -n 6 gpurun_out/r5n/pytest_a.txt
bash tools/gpu/sweep2.sh default head notail tb32 tb128 push128 push192 default head > gpurun_out/r5n/sweep.txt 2>&1; cat gpurun_out/r5n/sweep.txt
for v in head push128; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5n/share_$v.txt 2>&1; done
timeout 200 python tools/gpu/share.py > gpurun_out/r5n/share_default.txt 2>&1
tail -qn 1 gpurun_out/r5n/share_*.txt
