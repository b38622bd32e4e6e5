cd /root/repo
mkdir -p gpurun_out/r4d
for v in "$@"; do
  if [ "$v" == "default" ]; then L=""; else L="variants/$v"; fi
  LD_LIBRARY_PATH=$L timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "supergrid or majorant" > gpurun_out/r4d/t_$v.txt 2>&1
  echo "== $v"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4d/t_$v.txt | head -30
done
