# Usage: tools/gpu/sweep3.sh NAME...  -> headline at factor 8 and the envmap + factor-8 set-up per variant library
cd /root/repo
for v in "$@"; do
  if [ "$v" == "default" ]; then L=""; else L="variants/$v"; fi
  a=$(LD_LIBRARY_PATH=$L timeout 100 python bench.py --no-cpu-baseline --no-extra-configs --majorant-factor 8 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['t_primal_ms'], d['t_adjoint_ms'], d['t_grad_reduce_ms'])")
  b=$(LD_LIBRARY_PATH=$L timeout 200 python bench.py --only-config headline_envmap_factor8 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())['headline_envmap_factor8']; print(d['value'], d['t_primal_ms'], d['t_adjoint_ms'], d['t_grad_reduce_ms'])")
  echo "$v | headline8: $a | envmap8: $b"
done
