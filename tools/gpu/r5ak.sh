cd /root/repo; mkdir -p gpurun_out/r5ak
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r5ak/pytest_a.txt 2>&1; tail -n 3 gpurun_out/r5ak/pytest_a.txt
for v in default nosolo default nosolo; do
  if [ "$v" == "default" ]; then L=""; else L="variants/$v"; fi
  a=$(LD_LIBRARY_PATH=$L timeout 300 python bench.py --only-config config4_512_rank_share_1024x64 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())['config4_512_rank_share_1024x64']; print(d['value'], d['t_primal_ms'], d['t_adjoint_ms'], d['t_grad_reduce_ms'])")
  echo "$v | config4: $a"
done > gpurun_out/r5ak/c4.txt 2>&1
cat gpurun_out/r5ak/c4.txt
