cd /root/repo; mkdir -p gpurun_out/r5af
for v in default nosolo soloprimal soloadj default nosolo; do
  if [ "$v" == "default" ]; then L=""; else L="variants/$v"; fi
  a=$(LD_LIBRARY_PATH=$L timeout 300 python bench.py --only-config config3_optimize_loop 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())['config3_optimize_loop']; print(d['value'], d['global_majorant']['value'], d['envmap_factor8']['value'])")
  echo "$v | config3: $a"
done > gpurun_out/r5af/c3.txt 2>&1
cat gpurun_out/r5af/c3.txt
