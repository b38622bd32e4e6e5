#!/bin/bash
# Usage: tools/mk_variant_src.sh NAME path/to/alternative_drt_sq.hip ["-D..."]   -> variants/NAME/libdrt_hip.so with drt_sq.hip replaced by that file
set -e
name=$1; src=$2; defs=$3
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/unbiased-inverse-volume-rendering_amd/csrc
out=$root/variants/$name
mkdir -p $out
objs=()
for o in $csrc/_obj/*.o; do b=$(basename $o .o); [ "$b" != "drt_sq.hip" ] && objs+=($o); done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall $defs -I$csrc -c -x hip $src -o $out/drt_sq.hip.o
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" $out/drt_sq.hip.o -o $out/libdrt_hip.so
echo built $out/libdrt_hip.so
