cd /root/repo; R=gpurun_out/r05c; mkdir -p $R
(timeout 1500 python bench.py > $R/bench.json 2>> $R/err.txt)
(timeout 600 python bench.py --majorant-factor 0 --no-extra-configs --no-cpu-baseline > $R/bench_factor0.json 2>> $R/err.txt)
python - <<P
import json
for f in ("bench.json", "bench_factor0.json"):
    d=json.load(open("$R/" + f)); print(f, d["value"], d["t_primal_ms"], d["t_adjoint_ms"], d["t_grad_reduce_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], (d["roofline"].get("secondary") or {}).get("tile_reduce"))
d=json.load(open("$R/bench.json"))
print({k:(v.get("value"),v.get("error")) for k,v in d["other_configs"].items()})
print(d["cpu_baseline"])
P
