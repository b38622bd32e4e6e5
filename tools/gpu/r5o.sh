# round 5, call O: tail pool v3 with its size threshold: full GPU suite, bench lines
cd /root/repo
mkdir -p gpurun_out/r5o
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5o/pytest.txt 2>&1; tail -n 6 gpurun_out/r5o/pytest.txt
timeout 200 python tools/gpu/share.py > gpurun_out/r5o/share_default.txt 2>&1; tail -n 1 gpurun_out/r5o/share_default.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r5o/bench.json 2> gpurun_out/r5o/bench.err; python - <<'P'
import json
d = json.loads(open('/root/repo/gpurun_out/r5o/bench.json').read())
print(d['value'], d['ms_per_step'], d.get('roofline'))
for k, v in d.get('other_configs', {}).items():
    print(k, v.get('value'), v.get('unit'), v.get('ms_per_step'))
P
