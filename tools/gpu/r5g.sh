# round 5, call G: drained-mode latency variants (headline + rank shares), cache depth
cd /root/repo
mkdir -p gpurun_out/r5g
bash tools/gpu/sweep2.sh default di8 di32 dk4 di32dk4 cap16 default > gpurun_out/r5g/sweep.txt 2>&1
cat gpurun_out/r5g/sweep.txt
for v in di32 dk4 di32dk4 cap16; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5g/share_$v.txt 2>&1; done
timeout 200 python tools/gpu/share.py > gpurun_out/r5g/share_default.txt 2>&1
tail -qn 1 gpurun_out/r5g/share_*.txt
