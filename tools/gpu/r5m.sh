# round 5, call M: tail pool v2 (flag + break in the loop, cooperative hand-over behind it): parity first, then timing against HEAD's kernel
cd /root/repo
mkdir -p gpurun_out/r5m
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_estimators.py -m gpu -x -q > gpurun_out/r5m/pytest_a.txt 2>&1; tail -n 6 gpurun_out/r5m/pytest_a.txt
bash tools/gpu/sweep2.sh default head notail tb128 tb256 push32 push128 default head > gpurun_out/r5m/sweep.txt 2>&1; cat gpurun_out/r5m/sweep.txt
for v in head tb256; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5m/share_$v.txt 2>&1; done
timeout 200 python tools/gpu/share.py > gpurun_out/r5m/share_default.txt 2>&1
tail -qn 1 gpurun_out/r5m/share_*.txt
