cd /root/repo; mkdir -p gpurun_out/r5ai
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envmap.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r5ai/pytest_a.txt 2>&1; tail -n 4 gpurun_out/r5ai/pytest_a.txt
bash tools/gpu/sweep3.sh default nofinish default nofinish > gpurun_out/r5ai/sweep.txt 2>&1; cat gpurun_out/r5ai/sweep.txt
