# round 5, call W: full GPU suite with the nerf tile adjoint / fused split; bench side entries of config 5
cd /root/repo
mkdir -p gpurun_out/r5w
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_estimators.py --deselect tests/test_gpu_envmap.py > gpurun_out/r5w/pytest.txt 2>&1; tail -n 8 gpurun_out/r5w/pytest.txt
