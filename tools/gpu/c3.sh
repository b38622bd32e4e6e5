# Usage: tools/gpu/c3.sh NAME...  -> config 3 (optimisation loop, 200 iterations) it/s per variant library: factor 8 | global majorant | envmap + factor 8
cd /root/repo
for v in "$@"; do
  if [ "$v" == "default" ]; then L=""; else L="variants/$v"; fi
  LD_LIBRARY_PATH=$L timeout 300 python bench.py --only-config config3_optimize_loop 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())['config3_optimize_loop']; print('$v', d.get('value'), d.get('global_majorant',{}).get('value'), d.get('envmap_factor8',{}).get('value'), d.get('error'))"
done
