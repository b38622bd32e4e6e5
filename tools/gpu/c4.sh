# Usage: tools/gpu/c4.sh NAME...  -> config 4 (512^3, rank 0's share of 1024^2 x 64 spp) at factor 8 per variant library: Msamples/s, primal / adjoint / reductions ms
cd /root/repo
for v in "$@"; do
  if [ "$v" == "default" ]; then L=""; else L="variants/$v"; fi
  LD_LIBRARY_PATH=$L timeout 300 python bench.py --only-config config4_512_rank_share_1024x64 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())['config4_512_rank_share_1024x64']; print('$v', d.get('value'), d.get('t_primal_ms'), d.get('t_adjoint_ms'), d.get('t_grad_reduce_ms'), d.get('error'))"
done
