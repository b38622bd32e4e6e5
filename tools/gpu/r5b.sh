# round 5, call B: timing sweep of the first experiments (one box) + finish-age profiles of the order / urgency variants
cd /root/repo
mkdir -p gpurun_out/r5b
bash tools/gpu/sweep2.sh default nog4 ord0 ord4 cap64 cap64o0 g4 urg8 urg16 urg8np urg4f0 nt1 nt7 alb0 default nog4 > gpurun_out/r5b/sweep.txt 2>&1
cat gpurun_out/r5b/sweep.txt
LD_LIBRARY_PATH=variants/prof6 DRT_PROFILE_SPP=32 timeout 300 python tools/finish_age_profile.py > gpurun_out/r5b/finish_age_ord1.txt 2>&1
for v in urg8 cap64; do LD_LIBRARY_PATH=variants/$v timeout 200 python tools/gpu/share.py > gpurun_out/r5b/share_$v.txt 2>&1; done
timeout 200 python tools/gpu/share.py > gpurun_out/r5b/share_default.txt 2>&1
tail -n 2 gpurun_out/r5b/share_*.txt
