# Usage: tools/gpu/env8.sh NAME...  -> headline with a 2048x1024 environment map at factor 8 per variant library: Msamples/s, primal / adjoint / reductions ms
cd /root/repo
for v in "$@"; do
  if [ "$v" == "default" ]; then L=""; else L="variants/$v"; fi
  LD_LIBRARY_PATH=$L timeout 300 python bench.py --only-config headline_envmap_factor8 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())['headline_envmap_factor8']; print('$v', d.get('value'), d.get('t_primal_ms'), d.get('t_adjoint_ms'), d.get('t_grad_reduce_ms'), d.get('error'))"
done
