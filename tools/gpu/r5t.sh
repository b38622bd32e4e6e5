cd /root/repo; mkdir -p gpurun_out/r5v
timeout 1200 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_fused.py -m gpu -x -q > gpurun_out/r5v/pytest_a.txt 2>&1; tail -n 3 gpurun_out/r5v/pytest_a.txt
for c in config5_nerf_256_512x32 config5_fused_nerf_drt_256_512x32; do timeout 600 python bench.py --only-config $c > gpurun_out/r5v/$c.json 2> gpurun_out/r5v/$c.err; python - <<P
import json
d = json.loads(open('/root/repo/gpurun_out/r5v/$c.json').read())
d = d.get('$c', d)
print('$c', {k: d.get(k) for k in ('value', 'ms_per_step', 't_primal_ms', 't_adjoint_pass_ms', 'error', 'envmap_factor8')})
P
done
R=/root/repo/gpurun_out/r5v
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --only-config config5_nerf_256_512x32"
(timeout 600 rocprofv3 -i /root/repo/tools/pmc_lds.txt --kernel-trace --output-format csv -d $R/l -- $B > /dev/null 2> $R/err.txt)
cd /root/repo
python tools/pmc_summary.py $R/l > $R/pmc_lds.txt 2>&1
rm -rf $R/l
grep -A18 "nerf_tile" $R/pmc_lds.txt | head -20
