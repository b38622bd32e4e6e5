# round 5, call K: what the launches cost without the latency of their last paths (timing builds: drained workgroups drop their last records)
cd /root/repo
mkdir -p gpurun_out/r5k
bash tools/gpu/sweep2.sh default drop32 drop64 drop128 default > gpurun_out/r5k/sweep.txt 2>&1; cat gpurun_out/r5k/sweep.txt
for v in drop64; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5k/share_$v.txt 2>&1; done
tail -qn 1 gpurun_out/r5k/share_*.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r5k/pytest.txt 2>&1; tail -n 5 gpurun_out/r5k/pytest.txt
