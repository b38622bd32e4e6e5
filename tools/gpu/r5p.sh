# round 5, call P: tail pool v3 + barrier behind the tail launch's prologue: full GPU suite, stress, instruction-cache counters
cd /root/repo
mkdir -p gpurun_out/r5p
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5p/pytest.txt 2>&1; tail -n 6 gpurun_out/r5p/pytest.txt
timeout 900 python tools/stress_super.py --reps 30 > gpurun_out/r5p/stress.txt 2>&1; tail -n 6 gpurun_out/r5p/stress.txt
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_WAIT_INST[A-Z_0-9]*" | sort -u > gpurun_out/r5p/counters.txt; wc -l gpurun_out/r5p/counters.txt
R=/root/repo/gpurun_out/r5p
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --majorant-factor 8 --steps 3 --warmup 1"
(timeout 600 rocprofv3 -i /root/repo/tools/pmc_icache.txt --kernel-trace --output-format csv -d $R/ic -- $B > /dev/null 2> $R/err.txt)
(timeout 600 rocprofv3 -i /root/repo/tools/pmc_icache.txt --kernel-trace --output-format csv -d $R/ic_env -- python /root/repo/bench.py --only-config headline_envmap_factor8 > /dev/null 2>> $R/err.txt)
cd /root/repo
python tools/pmc_summary.py $R/ic > $R/pmc_icache.txt 2>&1
python tools/pmc_summary.py $R/ic_env > $R/pmc_icache_env.txt 2>&1
rm -rf $R/ic $R/ic_env
grep -A12 "trace_s" $R/pmc_icache.txt | head -60
tail -5 $R/err.txt
