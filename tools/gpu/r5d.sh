# round 5, call D: GPU test suite + A/B of the empty-pixel flags and the path cache depth
cd /root/repo
mkdir -p gpurun_out/r5d
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5d/pytest.txt 2>&1; tail -n 15 gpurun_out/r5d/pytest.txt
bash tools/gpu/sweep2.sh default noue cap64 default noue > gpurun_out/r5d/sweep.txt 2>&1
cat gpurun_out/r5d/sweep.txt
LD_LIBRARY_PATH=variants/prof6 DRT_PROFILE_SPP=32 timeout 300 python tools/finish_age_profile.py > gpurun_out/r5d/finish_age.txt 2>&1
for v in noue cap64; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5d/share_$v.txt 2>&1; done
timeout 200 python tools/gpu/share.py > gpurun_out/r5d/share_default.txt 2>&1
tail -qn 1 gpurun_out/r5d/share_*.txt
