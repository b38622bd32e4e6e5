cd /root/repo
mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "supergrid or majorant" > gpurun_out/r4j/t1.txt 2>&1; echo "rc $?" >> gpurun_out/r4j/t1.txt
tail -3 gpurun_out/r4j/t1.txt
bash tools/gpu/sweep2.sh "$@" 2>&1 | tee gpurun_out/r4j/sweep.txt
