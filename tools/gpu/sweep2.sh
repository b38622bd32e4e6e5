# Usage: tools/gpu/sweep2.sh NAME...  -> headline at factor 8 and config 2 (smoke, factor 8) per variant library
cd /root/repo
for v in "$@"; do
  if [ "$v" == "default" ]; then L=""; else L="variants/$v"; fi
  a=$(LD_LIBRARY_PATH=$L timeout 100 python bench.py --no-cpu-baseline --no-extra-configs --majorant-factor 8 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['t_primal_ms'], d['t_adjoint_ms'], d['t_grad_reduce_ms'])")
  b=$(LD_LIBRARY_PATH=$L timeout 100 python bench.py --no-cpu-baseline --no-extra-configs --majorant-factor 8 --workload smoke --res 128 --spp 16 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['t_primal_ms'], d['t_adjoint_ms'], d['t_grad_reduce_ms'])")
  echo "$v | headline8: $a | smoke8: $b"
done
