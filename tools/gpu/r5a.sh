# round 5, call A: baseline + cheap experiments (one box): timing sweep, finish-age profile at 32 and 4 spp
cd /root/repo
mkdir -p gpurun_out/r5a
tools/gpu/sweep2.sh default base nt1 nt2 nt7 alb0 base > gpurun_out/r5a/sweep.txt 2>&1
for spp in 32 4; do
  LD_LIBRARY_PATH=variants/prof6 DRT_PROFILE_SPP=$spp timeout 300 python tools/finish_age_profile.py >> gpurun_out/r5a/finish_age.txt 2>&1
done
LD_LIBRARY_PATH=variants/base timeout 200 python tools/gpu/share.py > gpurun_out/r5a/share.txt 2>&1
cat gpurun_out/r5a/sweep.txt gpurun_out/r5a/share.txt
