cd /root/repo
mkdir -p gpurun_out/r4b
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "supergrid or majorant" > gpurun_out/r4b/t1.txt 2>&1; echo "rc $?" >> gpurun_out/r4b/t1.txt
tail -15 gpurun_out/r4b/t1.txt
timeout 200 python bench.py --no-cpu-baseline --no-extra-configs --majorant-factor 8 --steps 10 --warmup 3 > gpurun_out/r4b/bench8.json 2> gpurun_out/r4b/bench8.err; echo "bench rc $?"
python - <<P
import json
for f in ("bench8",):
    try:
        d=json.load(open("gpurun_out/r4b/%s.json"%f)); print(f, d["value"], d["t_primal_ms"], d["t_adjoint_ms"], d["t_grad_reduce_ms"])
    except Exception as e: print(f, "failed", e)
P
tail -5 gpurun_out/r4b/bench8.err
