cd /root/repo; mkdir -p gpurun_out/r5al
timeout 1200 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_fused.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r5al/pytest_a.txt 2>&1; tail -n 3 gpurun_out/r5al/pytest_a.txt
for c in config5_nerf_256_512x32 config5_fused_nerf_drt_256_512x32; do timeout 600 python bench.py --only-config $c > gpurun_out/r5al/$c.json 2> gpurun_out/r5al/$c.err; python - <<P
import json
d = json.loads(open('/root/repo/gpurun_out/r5al/$c.json').read())
d = d.get('$c', d)
print('$c', {k: d.get(k) for k in ('value', 'ms_per_step', 't_primal_ms', 't_adjoint_pass_ms', 'error', 'envmap_factor8')})
P
done
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
