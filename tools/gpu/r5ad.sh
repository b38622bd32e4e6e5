cd /root/repo; mkdir -p gpurun_out/r5ad
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5ad/pytest_a.txt 2>&1; tail -n 3 gpurun_out/r5ad/pytest_a.txt
timeout 200 python tools/gpu/share.py > gpurun_out/r5ad/share_default.txt 2>&1
for v in push128; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5ad/share_$v.txt 2>&1; done
tail -qn 1 gpurun_out/r5ad/share_*.txt
