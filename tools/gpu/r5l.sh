# round 5, call L: tail pool of the queued tracer's adjoint launches (parity first, then timing)
cd /root/repo
mkdir -p gpurun_out/r5l
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_estimators.py -m gpu -x -q > gpurun_out/r5l/pytest_a.txt 2>&1; tail -n 6 gpurun_out/r5l/pytest_a.txt
bash tools/gpu/sweep2.sh default notail push32 push128 push256 default notail > gpurun_out/r5l/sweep.txt 2>&1; cat gpurun_out/r5l/sweep.txt
for v in notail push128; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5l/share_$v.txt 2>&1; done
timeout 200 python tools/gpu/share.py > gpurun_out/r5l/share_default.txt 2>&1
tail -qn 1 gpurun_out/r5l/share_*.txt
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r5l/pytest.txt 2>&1; tail -n 6 gpurun_out/r5l/pytest.txt
