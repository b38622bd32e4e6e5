cd /root/repo
tag=${1:-bench_full}
python bench.py ${@:2} > gpurun_out/$tag.json 2> gpurun_out/$tag.err; echo "rc $?"; tail -3 gpurun_out/$tag.err
python - <<P
import json
d=json.load(open("gpurun_out/$tag.json"))
print("main", d["value"], d["t_primal_ms"], d["t_adjoint_ms"], d["t_grad_reduce_ms"], "frac", d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
for k,v in (d["other_configs"] or {}).items():
    extra = {kk: vv.get("value") for kk, vv in v.items() if isinstance(vv, dict) and "value" in vv}
    print(k, v.get("value"), v.get("unit"), v.get("t_primal_ms"), v.get("t_adjoint_ms"), v.get("t_grad_reduce_ms"), (v.get("roofline") or {}).get("frac"), v.get("error"), extra)
P
