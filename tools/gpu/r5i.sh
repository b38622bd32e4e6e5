# round 5, call I: GPU suite after the clean-up (older tracers hooks-only), bench sanity
cd /root/repo
mkdir -p gpurun_out/r5i
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r5i/pytest.txt 2>&1; tail -n 25 gpurun_out/r5i/pytest.txt
bash tools/gpu/sweep2.sh default > gpurun_out/r5i/sweep.txt 2>&1; cat gpurun_out/r5i/sweep.txt
