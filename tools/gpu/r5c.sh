# round 5, call C: clean timing sweep (one box)
cd /root/repo
mkdir -p gpurun_out/r5c
bash tools/gpu/sweep2.sh default base ord0 nt1 nt2 nt4 nt7 nt8 nt15 nt23 nt31 urg8np urg12np urg16np urg8f30 urg8f70 urg8 cap32 cap64 combo1 combo2 default base > gpurun_out/r5c/sweep.txt 2>&1
cat gpurun_out/r5c/sweep.txt
for v in prof6 prof6u; do LD_LIBRARY_PATH=variants/$v DRT_PROFILE_SPP=32 timeout 300 python tools/finish_age_profile.py > gpurun_out/r5c/finish_age_$v.txt 2>&1; done
for v in base urg8np cap64 combo2; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5c/share_$v.txt 2>&1; done
tail -qn 1 gpurun_out/r5c/share_*.txt
