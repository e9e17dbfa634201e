#!/bin/bash
# One GPU-box visit that produces everything committed under profiles/ for a round:
#   GPU parity tests, the default bench line, a rocprofv3 kernel trace of the same step, PMC passes (FETCH_SIZE,
#   WRITE_SIZE, SQ wave cycles) over the whole-shard step, the small-batch phase counters, and a kernel trace + PMC
#   passes of the Hogwild engine.
# usage (repo root, on the GPU box): bash tools/final_run.sh <tag>
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
{ nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; rocm-smi --showproductname 2>/dev/null | head -8; } > $OUT/env.txt 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt
echo "== bench (default flags)"
timeout 900 python bench.py 2> $OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err; cut -c1-400 $OUT/bench.json
echo "== rocprofv3 kernel trace of the bench step"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep --no-parity-gate > $OUT/bench_prof.json 2> $OUT/bench_prof.err )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -8 "$f" | cut -c1-160
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python - "$f" > $OUT/dispatch_durations.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("per-kernel dispatch durations of the LAST 20 dispatches (us): name, n, mean, min, max")
for k, v in sorted(agg.items()):
    v = v[-20:]
    print("%-62s %4d %10.1f %10.1f %10.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
PY
rm -rf $OUT/prof
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  echo "== pmc pass $i: $P" | tee -a $OUT/pmc_summary.txt
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python $REPO/tools/pmc_step.py 8388608 3 > $OUT/pmc$i.out 2> $OUT/pmc$i.err )
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" "dsgd_" | grep -E "wseg_kernel<true|cdot|cgrad|cold|reduce|apply|bound" | tee -a $OUT/pmc_summary.txt
  rm -rf $OUT/pmc$i $OUT/pmc$i.out $OUT/pmc$i.err
done
echo "== small-batch phase counters and Hogwild by worker count"
timeout 300 python tools/plan_prof.py > $OUT/plan_prof.json 2> $OUT/plan_prof.err; head -c 600 $OUT/plan_prof.json
echo "== Hogwild: kernel trace and PMC"
timeout 300 python tools/hog_prof.py 8388608 256 60000 > $OUT/hogwild_8m.json 2> $OUT/hogwild_8m.err; cat $OUT/hogwild_8m.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hogtrace -o hog -- python $REPO/tools/hog_prof.py 8388608 256 60000 > /dev/null 2> $OUT/hog_trace.err )
f=$(find $OUT/hogtrace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|hogwild|eval" "$f" | cut -c1-200 | tee $OUT/hogwild_kernel_stats.csv
rm -rf $OUT/hogtrace
i=0
for P in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TA_BUSY_avr"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/hpmc$i -o pmc -- python $REPO/tools/hog_prof.py 8388608 256 60000 > /dev/null 2> $OUT/hpmc$i.err )
  f=$(find $OUT/hpmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "dsgd_hogwild" | tee -a $OUT/hogwild_pmc_summary.txt; else tail -5 $OUT/hpmc$i.err; fi
  rm -rf $OUT/hpmc$i $OUT/hpmc$i.err
done
