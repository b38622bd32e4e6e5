# round 5, call J: flagged-empty units last in the order; new tests
cd /root/repo
mkdir -p gpurun_out/r5j
bash tools/gpu/sweep2.sh default el0 el4 default el0 > gpurun_out/r5j/sweep.txt 2>&1; cat gpurun_out/r5j/sweep.txt
LD_LIBRARY_PATH=variants/prof6 DRT_PROFILE_SPP=32 timeout 300 python tools/finish_age_profile.py > gpurun_out/r5j/finish_age.txt 2>&1
for v in el0; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5j/share_$v.txt 2>&1; done
timeout 200 python tools/gpu/share.py > gpurun_out/r5j/share_default.txt 2>&1
tail -qn 1 gpurun_out/r5j/share_*.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_primitives.py -m gpu -x -q -s -k "reproduce or pack or config2 or headline" > gpurun_out/r5j/pytest.txt 2>&1; grep -n "grad_pack\|passed\|failed\|Error" gpurun_out/r5j/pytest.txt | tail -8
