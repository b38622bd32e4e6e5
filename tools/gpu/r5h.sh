# round 5, call H: envmap texel table (tests + bench entries), full GPU suite
cd /root/repo
mkdir -p gpurun_out/r5h
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5h/pytest.txt 2>&1; tail -n 6 gpurun_out/r5h/pytest.txt
for c in headline_envmap_factor8 headline_envmap; do timeout 600 python bench.py --only-config $c > gpurun_out/r5h/$c.json 2>/dev/null; python - <<P
import json
d=json.load(open("gpurun_out/r5h/$c.json")); k=list(d)[0]; r=d[k]
print(k, r.get("value"), r.get("t_primal_ms"), r.get("t_adjoint_ms"), r.get("t_grad_reduce_ms"), r.get("error"))
P
done
bash tools/gpu/sweep2.sh default > gpurun_out/r5h/sweep.txt 2>&1; cat gpurun_out/r5h/sweep.txt
