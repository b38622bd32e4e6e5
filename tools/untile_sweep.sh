#!/bin/bash
# sweep the gradient-reduction kernel's z-chunk; prints t_grad_reduce_ms
for res in 256 512; do
for z in 8 16 32 64 512; do
  for fl in 0 64; do
    echo -n "res $res Z $z flags $fl: "
    DRT_UNTILE_CHUNK=$z timeout -k 5 200 python bench.py --res $res --steps 4 --warmup 2 --no-cpu-baseline --debug-flags $fl 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['t_adjoint_ms'], d['t_grad_reduce_ms'])"
  done
done
done
