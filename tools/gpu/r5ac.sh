cd /root/repo; mkdir -p gpurun_out/r5ac
bash tools/gpu/sweep2.sh default push32 push128 push256 default push128 > gpurun_out/r5ac/sweep.txt 2>&1; cat gpurun_out/r5ac/sweep.txt
for v in push128 push256; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5ac/share_$v.txt 2>&1; done
timeout 200 python tools/gpu/share.py > gpurun_out/r5ac/share_default.txt 2>&1
tail -qn 1 gpurun_out/r5ac/share_*.txt
