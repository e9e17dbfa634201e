#!/bin/bash
# One GPU-box visit of round 5: pick the legs with arguments (the legs of tools/r04_visit.sh are available too).
# usage (repo root, on the GPU box): bash tools/r05_visit.sh <tag> [build] [testcs] [probe] [testall] [bench] [benchprof] [pmc] [latency] [cstrace] ...
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
for leg in "$@"; do
case $leg in
build)
  python -c "import __graft_entry__ as g; g.build()" > $OUT/build.txt 2>&1; tail -2 $OUT/build.txt ;;
testcs)
  echo "== the column-slice modules"
  timeout 900 python -m pytest tests/test_gpu_cs_device.py tests/test_gpu_cs.py -q -m gpu --tb=short -rf > $OUT/pytest_cs.txt 2>&1
  grep -E "^(E  |FAILED|ERROR)|passed|failed" $OUT/pytest_cs.txt | cut -c1-300 | head -60 ;;
probe)
  echo "== round-5 probe"
  timeout 900 python tools/r05_probe.py > $OUT/probe.json 2> $OUT/probe.err; tail -3 $OUT/probe.err
  python - <<PY
import json
d = json.load(open("$OUT/probe.json"))
for key in ("rows=23149", "rows=804414"):
    r = d[key]
    print(key, "fit", {k: round(v["batch_loop_us_per_step"], 2) for k, v in r["fit"].items()}, "exposed shuffle", {k: round(v["shuffle_us_per_step_exposed"], 2) for k, v in r["fit"].items()})
    print("   plan cycle (last)", {k: round(v, 3) if isinstance(v, float) else v for k, v in r["plan_cycle_3x100"][-1].items()})
    print("   requests", {m: {c: round(v["us_per_request_python_binding"], 1) for c, v in r[m].items()} for m in r if m.startswith("requests")})
    print("   with a communicator", r.get("with_communicator_3x100"))
PY
  ;;
testtc)
  echo "== the column-list module"
  timeout 900 python -m pytest tests/test_gpu_tcol.py -q -m gpu --tb=short -rf > $OUT/pytest_tcol.txt 2>&1
  grep -E "^(E  |FAILED|ERROR)|passed|failed" $OUT/pytest_tcol.txt | cut -c1-300 | head -60 ;;
tcprobe)
  echo "== column lists vs the row-wise kernel vs row chunks"
  timeout 900 python tools/tcol_probe.py ${TC_SIZES:-23149,100552} ${TC_VARIANTS:-rowwise,chunks,columns} > $OUT/tcol_probe.json 2> $OUT/tcol_probe.err
  cat $OUT/tcol_probe.err | tail -12 ;;
tcprof)
  echo "== column lists under rocprofv3 (kernel trace)"
  cd /tmp
  for sz in ${TC_PROF_SIZES:-23149 100552}; do
    for var in ${TC_PROF_VARIANTS:-columns}; do
      rm -rf /tmp/tcprof_$sz
      PROBE_WORKERS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tcprof_$sz -o tc -- python $REPO/tools/tcol_probe.py $sz $var > /dev/null 2> $OUT/tcprof_${sz}_$var.err
      f=$(find /tmp/tcprof_$sz -name "*kernel_stats.csv" | head -1)
      echo "-- $sz $var"
      [ -n "$f" ] && cp $f $OUT/tcol_kernel_stats_${sz}_$var.csv && grep -E "dsgd_tc_dot|dsgd_tc_grad|fix_reduce_apply|mb_grad" $f | awk -F'","' '{print substr($1,1,60), $2, $4}'
    done
  done
  cd $REPO ;;
tcpmc)
  # rocprofv3 --pmc passes over whole-split steps of N = 23,149 and 100,552 rows (the column lists' kernels)
  for sz in 23149 100552; do
    i=0
    for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
      i=$((i+1))
      echo "== $sz rows, pmc pass $i: $P" | tee -a $OUT/tcol_pmc_summary.txt
      ( cd /tmp && DSGD_TCOL_MAX=200000 timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/tcpmc$i -o pmc -- python $REPO/tools/pmc_step.py $sz 20 > $OUT/tcpmc$i.out 2> $OUT/tcpmc$i.err )
      f=$(find $OUT/tcpmc$i -name "*counter_collection.csv" | head -1)
      [ -n "$f" ] && python tools/pmc_summary.py "$f" "dsgd_" | grep -E "tc_dot|tc_grad|reduce_apply" | tee -a $OUT/tcol_pmc_summary.txt
      rm -rf $OUT/tcpmc$i $OUT/tcpmc$i.out $OUT/tcpmc$i.err
    done
  done ;;
latency5)
  echo "== per-request latency at the boundary (tools/boundary_latency.py): as built, and with the one-launch request kernel"
  timeout 300 python tools/boundary_latency.py 2000000 > $OUT/boundary_latency.json 2> $OUT/boundary_latency.err; tail -2 $OUT/boundary_latency.err; cat $OUT/boundary_latency.json
  echo "-- DSGD_CS_REQ=1"
  DSGD_CS_REQ=1 timeout 300 python tools/boundary_latency.py 2000000 > $OUT/boundary_latency_cs_req.json 2>> $OUT/boundary_latency.err; grep -E "sync_step|batch" $OUT/boundary_latency_cs_req.json ;;
smoke)
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt ;;
*)
  bash tools/r04_visit.sh $TAG $leg ;;
esac
done
