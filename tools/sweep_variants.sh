#!/bin/bash
# Usage (GPU box, repo root): tools/sweep_variants.sh "<bench args>" NAME...   (NAME = a directory under variants/, or `default`)
# Runs bench.py once per variant library (built here by tools/mk_variant.sh) and prints value / primal / adjoint tracer / reductions.
bargs=$1; shift
for v in "$@"; do
  if [ "$v" == "default" ]; then L=""; else L="variants/$v"; fi
  LD_LIBRARY_PATH=$L python bench.py --no-cpu-baseline --no-extra-configs $bargs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', d['value'], d['t_primal_ms'], d['t_adjoint_ms'], d['t_grad_reduce_ms'])"
done
