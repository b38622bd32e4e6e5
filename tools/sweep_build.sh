#!/bin/bash
# Usage (on a GPU box, from the repo root): tools/sweep_build.sh "DRT_X=1 DRT_Y=2" "DRT_X=3" ... [-- bench args]
# Rebuilds the in-tree libraries with each set of -D overrides (numeric DRT_* environment variables, _build.py), runs
# bench.py, prints value / primal / adjoint tracer / reductions, and restores the default build at the end.
args=()
cfgs=()
seen=0
for a in "$@"; do
  if [ "$a" == "--" ]; then seen=1; continue; fi
  if [ $seen == 1 ]; then args+=("$a"); else cfgs+=("$a"); fi
done
for cfg in "${cfgs[@]}"; do
  env $cfg python -c "import __graft_entry__ as g; g.build()" > /tmp/sweep_build.log 2>&1 || { echo "build failed: $cfg"; tail -5 /tmp/sweep_build.log; continue; }
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --no-extra-configs "${args[@]}" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', d['value'], d['t_primal_ms'], d['t_adjoint_ms'], d['t_grad_reduce_ms'])"
  done
done
python -c "import __graft_entry__ as g; g.build()" > /tmp/sweep_build.log 2>&1 && echo "default build restored"
