#!/bin/bash
# Usage (GPU box, repo root): tools/gpu/pmc8.sh <tag> [bench args...]  -> utilisation + LDS counter passes of the factor-8 headline, tracer kernels only
tag=${1:-pmc8}; shift
R=/root/repo/gpurun_out/$tag
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --majorant-factor 8 --steps 3 --warmup 1 $@"
(timeout 600 rocprofv3 -i /root/repo/tools/pmc_util.txt --kernel-trace --output-format csv -d $R/u -- $B > /dev/null 2> $R/err.txt)
(timeout 600 rocprofv3 -i /root/repo/tools/pmc_lds.txt --kernel-trace --output-format csv -d $R/l -- $B > /dev/null 2>> $R/err.txt)
cd /root/repo
python tools/pmc_summary.py $R/u > $R/pmc_util.txt
python tools/pmc_summary.py $R/l > $R/pmc_lds.txt
rm -rf $R/u $R/l
grep -A19 "trace_s" $R/pmc_util.txt | grep -v "^--"
grep -A17 "trace_s" $R/pmc_lds.txt | grep -v "^--"
