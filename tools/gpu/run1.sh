cd /root/repo
mkdir -p gpurun_out/r4a
for m in 4 2 1; do DRT_PROFILE_MODE=$m LD_LIBRARY_PATH=variants/prof$m timeout 300 python tools/super_profile.py > gpurun_out/r4a/prof$m.txt 2>&1; done
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --majorant-factor 8 --steps 10 --warmup 3 > gpurun_out/r4a/bench8.json 2> gpurun_out/r4a/bench8.err
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 10 --warmup 3 > gpurun_out/r4a/bench0.json 2> gpurun_out/r4a/bench0.err
tail -3 gpurun_out/r4a/prof*.txt
python - <<P
import json
for f in ("bench8","bench0"):
    d=json.load(open("gpurun_out/r4a/%s.json"%f)); print(f, d["value"], d["t_primal_ms"], d["t_adjoint_ms"], d["t_grad_reduce_ms"])
P
