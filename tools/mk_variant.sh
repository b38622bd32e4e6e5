#!/bin/bash
# Usage (here, no GPU needed): tools/mk_variant.sh NAME "-DDRT_SUPER_K=4 ..." [source.hip ...]
# Builds variants/NAME/libdrt_hip.so: the production objects of csrc/_obj with the listed translation units (default:
# drt_super.hip) recompiled with the extra -D flags.  On the GPU box: LD_LIBRARY_PATH=variants/NAME python bench.py ...
# (the pybind shim finds libdrt_hip.so through RUNPATH, which LD_LIBRARY_PATH precedes).  variants/ is git-ignored.
set -e
name=$1; defs=$2; shift 2 || true
srcs=("$@"); [ ${#srcs[@]} -eq 0 ] && srcs=(drt_sq.hip)
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/unbiased-inverse-volume-rendering_amd/csrc
out=$root/variants/$name
mkdir -p $out
objs=()
for o in $csrc/_obj/*.o; do
  b=$(basename $o .o); skip=0
  for s in "${srcs[@]}"; do [ "$b" == "$s" ] && skip=1; done
  [ $skip == 0 ] && objs+=($o)
done
for s in "${srcs[@]}"; do
  unit=""; [ "$s" == "drt_sq.hip" ] && unit="-O2"          # (as _build.py: UNIT_FLAGS)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -DDRT_EXPERIMENT_BUILD=1 $unit $defs -c $csrc/$s -o $out/$s.o
  objs+=($out/$s.o)
done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o $out/libdrt_hip.so
echo "$name: $defs" > $out/flags.txt
echo built $out/libdrt_hip.so
