# round 5, call E: GPU test suite + A/B of the trivial-ray streaming pass
cd /root/repo
mkdir -p gpurun_out/r5e
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5e/pytest.txt 2>&1; tail -n 15 gpurun_out/r5e/pytest.txt
bash tools/gpu/sweep2.sh default noue notriv cap64 default noue > gpurun_out/r5e/sweep.txt 2>&1
cat gpurun_out/r5e/sweep.txt
LD_LIBRARY_PATH=variants/prof6 DRT_PROFILE_SPP=32 timeout 300 python tools/finish_age_profile.py > gpurun_out/r5e/finish_age.txt 2>&1
for v in noue cap64; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5e/share_$v.txt 2>&1; done
timeout 200 python tools/gpu/share.py > gpurun_out/r5e/share_default.txt 2>&1
tail -qn 1 gpurun_out/r5e/share_*.txt
