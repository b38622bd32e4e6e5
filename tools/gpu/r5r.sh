cd /root/repo; mkdir -p gpurun_out/r5r
timeout 1200 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_fused.py -m gpu -x -q > gpurun_out/r5r/pytest_a.txt 2>&1; tail -n 5 gpurun_out/r5r/pytest_a.txt
LD_LIBRARY_PATH=variants/ntstats timeout 300 python tools/gpu/nt_stats.py > gpurun_out/r5r/stats.txt 2>&1; tail -1 gpurun_out/r5r/stats.txt
for c in config5_nerf_256_512x32 config5_fused_nerf_drt_256_512x32; do timeout 600 python bench.py --only-config $c > gpurun_out/r5r/$c.json 2> gpurun_out/r5r/$c.err; python - <<P
import json
d = json.loads(open('/root/repo/gpurun_out/r5r/$c.json').read())
d = d.get('$c', d)
print('$c', {k: d.get(k) for k in ('value', 'ms_per_step', 't_primal_ms', 't_adjoint_pass_ms', 'error', 'envmap_factor8')})
P
done
