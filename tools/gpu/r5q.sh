# round 5, call Q: nerf adjoint with LDS pre-reduction (drt_nerf_tile.hip), fused pass as two dense passes
cd /root/repo
mkdir -p gpurun_out/r5q
timeout 1200 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_fused.py -m gpu -x -q > gpurun_out/r5q/pytest_a.txt 2>&1; tail -n 15 gpurun_out/r5q/pytest_a.txt
for c in config5_nerf_256_512x32 config5_fused_nerf_drt_256_512x32; do timeout 600 python bench.py --only-config $c > gpurun_out/r5q/$c.json 2> gpurun_out/r5q/$c.err; python - <<P
import json
d = json.loads(open('/root/repo/gpurun_out/r5q/$c.json').read())
d = d.get('$c', d)
print('$c', {k: d.get(k) for k in ('value', 'ms_per_step', 't_primal_ms', 't_adjoint_pass_ms', 'error', 'envmap_factor8')})
P
done
