cd /root/repo
mkdir -p gpurun_out/r4c
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "supergrid or majorant" > gpurun_out/r4c/t1.txt 2>&1; echo "rc $?" >> gpurun_out/r4c/t1.txt
tail -5 gpurun_out/r4c/t1.txt
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --majorant-factor 8 --steps 10 --warmup 3 > gpurun_out/r4c/bench8.json 2> gpurun_out/r4c/bench8.err; echo "bench rc $?"
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 10 --warmup 3 > gpurun_out/r4c/bench0.json 2> gpurun_out/r4c/bench0.err; echo "bench rc $?"
LD_LIBRARY_PATH=variants/sqprof timeout 300 python tools/super_profile.py > gpurun_out/r4c/sqprof.txt 2>&1
tail -4 gpurun_out/r4c/sqprof.txt
python - <<P
import json
for f in ("bench8","bench0"):
    try:
        d=json.load(open("gpurun_out/r4c/%s.json"%f)); print(f, d["value"], d["t_primal_ms"], d["t_adjoint_ms"], d["t_grad_reduce_ms"])
    except Exception as e: print(f, "failed", e)
P
