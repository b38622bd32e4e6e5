#!/bin/bash
# ONE parametrised runner for a gpurun call (round 6; the per-call scratch scripts of round 5 are gone):
#   gpurun --timeout N -- 'bash tools/gpu/run.sh TAG STEP [STEP ...]'        outputs -> gpurun_out/TAG/
# STEPs (in the order given):
#   tests[:EXPR]        pytest -m gpu (-k EXPR), tail of the log
#   vtests:V:EXPR       pytest -m gpu -k EXPR with LD_LIBRARY_PATH=variants/V (the production library only: hooks tests keep the in-tree one)
#   smoke               __graft_entry__.smoke()
#   bench               headline line (no side configs, no CPU baseline)
#   only:NAME           bench.py --only-config NAME
#   sweep:V1,V2,..      tools/gpu/sweep2.sh over variant libraries (variants/V/libdrt_hip.so; "default" = the in-tree library)
#   sweep3:V1,..        headline + config3_as_reproduce per variant
#   share:V1,..         a rank's share at G = 1 / 4 / 8 per variant
#   sqprof              block budget of the queued tracer: variants sqprof1..4 through tools/super_profile.py
#   c3prof:R1,R2[@V]    kernel trace of config 3's level R (tools/config3_level_profile.py), untrained and trained, library of variant V
#   traffic:V1,V2       counter traffic of the headline's kernels per variant
#   c3lvl:V1,V2,..      config 3's levels 16^3 / 256^3, untrained / trained, per variant, WITHOUT a profiler: it/s and the host's ms per iteration
#   config4ar           config 4's 2 GiB gradient exchange on one rank of RCCL: support / pack / collective / unpack timings
#   stress              tools/stress_super.py --reps 20
#   fuzz:LO:HI          tests/test_gpu_fuzz.py over the seeds LO .. HI-1 (random scenes against the oracle; no -x: every failing seed is listed)
TAG=$1; shift
cd /root/repo
R=/root/repo/gpurun_out/$TAG
mkdir -p $R
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "=== $step"
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout 2700 python -m pytest tests -m gpu -x -q -k "$arg" > $R/pytest.txt 2>&1; else timeout 2700 python -m pytest tests -m gpu -x -q > $R/pytest.txt 2>&1; fi
      tail -n 15 $R/pytest.txt ;;
    vtests)    # vtests:VARIANT:EXPR - the production-flavour tests matching EXPR against variants/VARIANT/libdrt_hip.so
      v=${arg%%:*}; ex=${arg#*:}
      LD_LIBRARY_PATH=variants/$v timeout 2400 python -m pytest tests -m gpu -x -q -k "$ex" > $R/pytest_$v.txt 2>&1; tail -n 6 $R/pytest_$v.txt ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 4 ;;
    bench) timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > $R/bench_headline.json 2> $R/bench_err.txt
           python -c "import json; d=json.load(open('$R/bench_headline.json')); print(d['value'], d['t_primal_ms'], d['t_adjoint_ms'], d['t_grad_reduce_ms'], d['roofline']['frac'])" ;;
    only) timeout 1200 python bench.py --only-config $arg > $R/only_$arg.json 2>> $R/bench_err.txt
          python -c "import json; d=json.load(open('$R/only_$arg.json')); print(json.dumps(d)[:1500])" ;;
    sweep) bash tools/gpu/sweep2.sh ${arg//,/ } | tee -a $R/sweep.txt ;;
    sweep3)
      for v in ${arg//,/ }; do
        L=""; [ "$v" != "default" ] && L="variants/$v"
        a=$(LD_LIBRARY_PATH=$L timeout 100 python bench.py --no-cpu-baseline --no-extra-configs --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['t_primal_ms'], d['t_adjoint_ms'], d['t_grad_reduce_ms'])")
        b=$(LD_LIBRARY_PATH=$L timeout 600 python bench.py --only-config config3_as_reproduce 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['config3_as_reproduce']; print(d.get('value'), [l['iterations_per_s'] for l in d.get('levels', [])], d.get('error'))")
        echo "$v | headline8: $a | config3_as_reproduce: $b" | tee -a $R/sweep3.txt
      done ;;
    share) for v in ${arg//,/ }; do L=""; [ "$v" != "default" ] && L="variants/$v"; echo "$v | $(LD_LIBRARY_PATH=$L timeout 300 python tools/gpu/share.py 2>/dev/null)" | tee -a $R/share.txt; done ;;
    sqprof)
      for m in 1 2 3 4; do
        mode=sq; [ $m != 1 ] && mode=sq$m
        echo "--- DRT_SQ_PROFILE=$m" >> $R/sqprof.txt
        LD_LIBRARY_PATH=variants/sqprof$m DRT_PROFILE_MODE=$mode timeout 300 python tools/super_profile.py >> $R/sqprof.txt 2>> $R/sqprof_err.txt
      done
      cat $R/sqprof.txt ;;
    c3prof)    # c3prof:RES[,RES..][@VARIANT]
      v=""; [[ "$arg" == *@* ]] && v=${arg#*@}; arg=${arg%%@*}
      L=""; [ -n "$v" ] && L="/root/repo/variants/$v"
      for res in ${arg//,/ }; do
        for tr in "" "--trained"; do
          sfx=$res; [ -n "$tr" ] && sfx=${res}_trained; [ -n "$v" ] && sfx=${sfx}_$v
          (cd /tmp && export TMPDIR=/tmp && LD_LIBRARY_PATH=$L timeout 900 rocprofv3 --kernel-trace --stats -d $R/c3_$sfx -o lv -- python /root/repo/tools/config3_level_profile.py --res $res --iters 40 $tr > $R/c3_$sfx.json 2>> $R/c3_err.txt)
          python tools/rocpd_stats.py $R/c3_$sfx/lv_results.db --csv $R/c3_${sfx}_kernel_stats.csv --top 12 > $R/c3_${sfx}_top.txt 2>&1
          rm -rf $R/c3_$sfx
          echo "--- $sfx"; python - $R/c3_${sfx}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:7.3f} min {float(r['MinNs'])/1e6:7.3f} max {float(r['MaxNs'])/1e6:7.3f} total {float(r['TotalDurationNs'])/1e6:8.1f}")
PY
        done
      done ;;
    traffic)   # traffic:V1,V2 - HBM-side counter traffic of the headline's kernels per variant (tools/pmc_traffic.txt, tools/pmc_to_traffic.py)
      for v in ${arg//,/ }; do
        L=""; [ "$v" != "default" ] && L="/root/repo/variants/$v"
        (cd /tmp && export TMPDIR=/tmp && LD_LIBRARY_PATH=$L timeout 900 rocprofv3 -i /root/repo/tools/pmc_traffic.txt --kernel-trace --output-format csv -d $R/pmc_traffic_$v -- python /root/repo/bench.py --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1 > /dev/null 2>> $R/traffic_err.txt)
        python tools/pmc_to_traffic.py $R/pmc_traffic_$v dust-devil-256-512x32-factor8 $R/traffic_$v.json > $R/pmc_traffic_$v.txt 2>&1
        rm -rf $R/pmc_traffic_$v
        echo "--- $v"; head -n 14 $R/pmc_traffic_$v.txt
      done ;;
    c3lvl)
      for v in ${arg//,/ }; do
        L=""; [ "$v" != "default" ] && L="variants/$v"
        for res in 16 256; do for tr in "" "--trained"; do
          echo "$v | $(LD_LIBRARY_PATH=$L timeout 300 python tools/config3_level_profile.py --res $res --iters 60 $tr 2>/dev/null)" | tee -a $R/c3lvl.txt
        done; done
      done ;;
    config4ar) HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 tests/workers/nccl_config4_worker.py 2> $R/config4ar_err.txt | tee $R/config4ar.txt ;;
    fuzz) DRT_FUZZ_SEEDS=$arg timeout 2700 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 6 > $R/fuzz.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $R/fuzz.txt | cut -c1-400 | tail -n 40 ;;
    stress) timeout 900 python tools/stress_super.py --reps 20 > $R/stress.txt 2>&1; tail -n 5 $R/stress.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
