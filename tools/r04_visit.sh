#!/bin/bash
# One GPU-box visit of round 4: pick the legs with arguments.
# usage (repo root, on the GPU box): bash tools/r04_visit.sh <tag> [test] [mb] [mbprof] [bench] [benchprof] [pmc] [hog]
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
{ nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; rocm-smi --showproductname 2>/dev/null | head -8; } > $OUT/env.txt 2>&1
trace_summary() {   # <kernel_trace.csv> -> per-kernel durations of the last 20 dispatches
python - "$1" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("per-kernel dispatch durations of the LAST 20 dispatches (us): name, n, mean, min, max")
for k, v in sorted(agg.items()):
    n = len(v); v = v[-20:]
    print("%-62s %5d %10.1f %10.1f %10.1f" % (k, n, sum(v) / len(v), min(v), max(v)))
PY
}
for leg in "$@"; do
case $leg in
test)
  echo "== pytest -m gpu"
  timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt ;;
testall)
  echo "== pytest -m gpu (no -x)"
  timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --tb=short -rf > $OUT/pytest_gpu_full.txt 2>&1
  grep -E "^(E  |FAILED|ERROR|tests/.*(Error|assert))|passed|failed" $OUT/pytest_gpu_full.txt | cut -c1-400 | head -80
  tail -45 $OUT/pytest_gpu_full.txt | cut -c1-300 | tee $OUT/pytest_gpu.txt ;;
testsel)
  echo "== pytest -m gpu -k \"$TESTSEL\""
  timeout 1800 python -m pytest tests -m gpu -q --timeout 900 --tb=short -rf -s -k "$TESTSEL" > $OUT/pytest_sel.txt 2>&1
  grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" $OUT/pytest_sel.txt | cut -c1-600 | tail -120 ;;
mb)
  echo "== index-list batches (tools/mb_prof.py)"
  timeout 600 python tools/mb_prof.py 2000000 > $OUT/mb_prof.json 2> $OUT/mb_prof.err; tail -3 $OUT/mb_prof.err; python -c "
import json; d=json.load(open('$OUT/mb_prof.json'))
for s in d['steps']: print('%d x %6d: %7.1f us/step %7.1f GB/s (%.3f of 8 TB/s) %s shift %d' % (s['workers'], s['batch'], s['us_per_step'], s['GBps'], s['frac_of_8TBps'], s['kernel'], s['fix_shift']))" ;;
mbcyc)
  echo "== phase cycles of wave 0 (DSGD_PLAN_PROF=1)"
  DSGD_PLAN_PROF=1 timeout 300 python tools/mb_prof.py 2000000 --only=3x100 --only=4x200 --only=1x100 > $OUT/mb_cycles.json 2> $OUT/mb_cycles.err; tail -2 $OUT/mb_cycles.err; python -c "
import json; d=json.load(open('$OUT/mb_cycles.json'))
for s in d['steps']: print(s['workers'], s['batch'], round(s['us_per_step'],1), {k: int(v) for k, v in s.get('wave0_cycles_per_launch', {}).items()})" ;;
mbcyc16)
  echo "== phase cycles, sixteen column slices (DSGD_CS_G=16)"
  DSGD_CS_G=16 DSGD_PLAN_PROF=1 timeout 300 python tools/mb_prof.py 2000000 --only=3x100 --only=4x200 --only=1x100 > $OUT/mb_cycles16.json 2> $OUT/mb_cycles16.err; tail -2 $OUT/mb_cycles16.err; python -c "
import json; d=json.load(open('$OUT/mb_cycles16.json'))
for s in d['steps']: print(s['workers'], s['batch'], round(s['us_per_step'],1), {k: int(v) for k, v in s.get('wave0_cycles_per_launch', {}).items()})" ;;
mbcyc1)
  echo "== phase cycles, ONE step per launch"
  DSGD_PLAN_PROF=1 timeout 300 python tools/mb_prof.py 2000000 --single --only=3x100 --only=4x200 > $OUT/mb_cycles1.json 2> $OUT/mb_cycles1.err; tail -2 $OUT/mb_cycles1.err; python -c "
import json; d=json.load(open('$OUT/mb_cycles1.json'))
for s in d['steps']: print(s['workers'], s['batch'], round(s['us_per_step'],1), {k: int(v) for k, v in s.get('wave0_cycles_per_launch', {}).items()})" ;;
mbcyc256)
  echo "== phase cycles, 256 lanes per slice (DSGD_CS_NT=256)"
  DSGD_CS_NT=256 DSGD_PLAN_PROF=1 timeout 300 python tools/mb_prof.py 2000000 --only=3x100 --only=1x100 > $OUT/mb_cycles256.json 2> $OUT/mb_cycles256.err; tail -2 $OUT/mb_cycles256.err; python -c "
import json; d=json.load(open('$OUT/mb_cycles256.json'))
for s in d['steps']: print(s['workers'], s['batch'], round(s['us_per_step'],1), {k: int(v) for k, v in s.get('wave0_cycles_per_launch', {}).items()})" ;;
mbtrace)
  echo "== rocprofv3 kernel trace of the index-list kernels, one batch size per run"
  for C in 1x65536 1x4096 3x100; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mbtrace -o mb -- python $REPO/tools/mb_prof.py 2000000 --only=$C > /dev/null 2> $OUT/mbtrace.err )
    f=$(find $OUT/mbtrace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && { echo "-- $C"; trace_summary "$f" | grep -E "mb_grad|vt_grad|fix_reduce"; } | tee -a $OUT/mb_dispatch_durations.txt
    f=$(find $OUT/mbtrace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "-- $C"; grep -E "Name|mb_grad|vt_grad|fix_reduce" "$f" | cut -c1-200; } >> $OUT/mb_kernel_stats.csv
    rm -rf $OUT/mbtrace
  done ;;
mbprof)
  echo "== rocprofv3 kernel trace + PMC of the index-list kernels"
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mbtrace -o mb -- python $REPO/tools/mb_prof.py 2000000 --quick > /dev/null 2> $OUT/mbtrace.err )
  f=$(find $OUT/mbtrace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|mb_grad|vt_grad|fix_reduce|plan_kernel" "$f" | cut -c1-220 | tee $OUT/mb_kernel_stats.csv
  f=$(find $OUT/mbtrace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && trace_summary "$f" | grep -E "per-kernel|mb_grad|vt_grad|fix_reduce" | tee $OUT/mb_dispatch_durations.txt
  rm -rf $OUT/mbtrace
  i=0
  for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TA_BUSY_avr"; do
    i=$((i+1))
    echo "== mb pmc pass $i: $P" | tee -a $OUT/mb_pmc_summary.txt
    ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/mbpmc$i -o pmc -- python $REPO/tools/mb_prof.py 2000000 --quick > /dev/null 2> $OUT/mbpmc$i.err )
    f=$(find $OUT/mbpmc$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "dsgd_" | grep -E "mb_grad|vt_grad|fix_reduce" | tee -a $OUT/mb_pmc_summary.txt; else tail -3 $OUT/mbpmc$i.err; fi
    rm -rf $OUT/mbpmc$i $OUT/mbpmc$i.err
  done ;;
bench)
  echo "== bench (default flags)"
  timeout 900 python bench.py 2> $OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err; cut -c1-600 $OUT/bench.json
  cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null ;;   # (the next bench run of any kind overwrites it)
benchquick)
  echo "== bench (no cpu baseline, no sweep, no gate)"
  timeout 600 python bench.py --no-cpu-baseline --no-sweep --no-parity-gate 2> $OUT/benchq.err > $OUT/benchq.json; tail -3 $OUT/benchq.err; cut -c1-900 $OUT/benchq.json ;;
benchprof)
  echo "== rocprofv3 kernel trace of the bench step"
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep --no-parity-gate --detail $OUT/bench_prof_detail.json > $OUT/bench_prof.json 2> $OUT/bench_prof.err )
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -8 "$f" | cut -c1-160
  f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && trace_summary "$f" > $OUT/dispatch_durations.txt && grep -E "fstep|wseg_kernel<true|cdot|cgrad|cold|reduce" $OUT/dispatch_durations.txt
  rm -rf $OUT/prof ;;
pmc)
  i=0
  for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"; do
    i=$((i+1))
    echo "== pmc pass $i: $P" | tee -a $OUT/pmc_summary.txt
    ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python $REPO/tools/pmc_step.py 8388608 3 > $OUT/pmc$i.out 2> $OUT/pmc$i.err )
    f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" "dsgd_" | grep -E "fstep|wseg_kernel<true|cdot|cgrad|cold|reduce|apply|bound" | tee -a $OUT/pmc_summary.txt
    rm -rf $OUT/pmc$i $OUT/pmc$i.out $OUT/pmc$i.err
  done ;;
streammin)
  echo "== whole-shard step at N = 23,149 / 100,000 by DSGD_STREAM_MIN (rows from which row ranges take the streaming kernels)"
  for M in 8192 32768 131072; do
    DSGD_STREAM_MIN=$M timeout 300 python -c "
import sys, json; sys.path.insert(0, '.')
import bench, dsgd_amd
bench.SWEEP = ()
for n in (23149, 100000):
    r = bench.reference_shape(dsgd_amd, 0, n, with_parity=True, repeats=5, steps=20)
    print('stream_min', $M, 'rows', n, round(r['whole_shard']['us_per_step'], 1), 'us', r['whole_shard']['kernel'], 'gate', r['parity_gate']['max_rel_err'])
" 2>&1 | grep -E "stream_min|Error|error" | tee -a $OUT/streammin.txt
  done ;;
latency)
  echo "== per-request latency at the boundary (tools/boundary_latency.py)"
  timeout 300 python tools/boundary_latency.py 2000000 > $OUT/boundary_latency.json 2> $OUT/boundary_latency.err; tail -2 $OUT/boundary_latency.err; cat $OUT/boundary_latency.json
  echo "-- with the copy on the stream (DSGD_REQ_MAPPED=0)"
  DSGD_REQ_MAPPED=0 timeout 300 python tools/boundary_latency.py 2000000 > $OUT/boundary_latency_copy.json 2>> $OUT/boundary_latency.err; grep -E "sync_step|batch" $OUT/boundary_latency_copy.json
  echo "-- waiting for the stream instead of polling the mailbox (DSGD_REQ_SPIN=0)"
  DSGD_REQ_SPIN=0 timeout 300 python tools/boundary_latency.py 2000000 > $OUT/boundary_latency_nospin.json 2>> $OUT/boundary_latency.err; grep -E "sync_step|batch" $OUT/boundary_latency_nospin.json
  echo "-- one-worker requests through the row-parallel kernels (DSGD_REQ_PLAN=0)"
  DSGD_REQ_PLAN=0 timeout 300 python tools/boundary_latency.py 2000000 > $OUT/boundary_latency_noplan.json 2>> $OUT/boundary_latency.err; grep -E "sync_step|batch" $OUT/boundary_latency_noplan.json ;;
cstrace)
  echo "== rocprofv3 kernel trace of the column-slice launches, one configuration per run"
  rm -f $OUT/cs_dispatch_durations.txt $OUT/cs_kernel_stats.csv
  for C in 3x100 4x200 1x100 "3x100 --single" "4x200 --single"; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cstrace -o cs -- python $REPO/tools/mb_prof.py 2000000 --only=$C > "$OUT/cstrace_$C.json" 2> $OUT/cstrace.err )
    f=$(find $OUT/cstrace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && { echo "-- $C"; trace_summary "$f" | grep -E "per-kernel|cs_step|vt_grad|fix_reduce|plan_kernel"; python -c "
import json; d=json.load(open('$OUT/cstrace_$C.json'))
for s in d['steps']: print('   steps per launch', s.get('steps_per_launch'), ' us/step (host clock)', round(s['us_per_step'], 2), s['kernel'])"; } | tee -a $OUT/cs_dispatch_durations.txt
    f=$(find $OUT/cstrace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "-- $C"; grep -E "Name|cs_step|vt_grad|fix_reduce|plan_kernel" "$f" | cut -c1-200; } >> $OUT/cs_kernel_stats.csv
    rm -rf $OUT/cstrace
  done ;;
variants)
  echo "== A/B variants of the whole-shard step ($VARIANTS)"
  eval "timeout 900 python tools/variants.py $VARIANTS" 2>&1 | tee $OUT/variants.txt ;;
hogcyc)
  echo "== Hogwild: cycles of thread 0 of worker 0 per iteration by phase (DSGD_PLAN_PROF=1), 256 / 64 / 1 workers"
  for W in 256 64 1; do
    DSGD_PLAN_PROF=1 timeout 300 python tools/hog_prof.py 2000000 $W $((W * 150 + 300)) > $OUT/hog_cycles_$W.json 2> $OUT/hog_cycles.err; python -c "
import json; d=json.load(open('$OUT/hog_cycles_$W.json'))
r=d['runs'][-1]; print(d['workers'], 'workers', round(r['us_per_iteration_per_worker'],1), 'us/iteration', round(r['examples_per_s']/1e6,1), 'M ex/s', {k:int(v) for k,v in r.get('worker0_cycles_per_iteration',{}).items()})"
  done ;;
hog)
  timeout 300 python tools/hog_prof.py 8388608 256 60000 > $OUT/hogwild_8m.json 2> $OUT/hogwild_8m.err; cat $OUT/hogwild_8m.json ;;
hogprof)
  echo "== Hogwild: kernel trace and PMC"
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hogtrace -o hog -- python $REPO/tools/hog_prof.py 8388608 256 60000 > /dev/null 2> $OUT/hog_trace.err )
  f=$(find $OUT/hogtrace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|hogwild|eval" "$f" | cut -c1-200 | tee $OUT/hogwild_kernel_stats.csv
  rm -rf $OUT/hogtrace
  i=0
  for P in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/hpmc$i -o pmc -- python $REPO/tools/hog_prof.py 8388608 256 60000 > /dev/null 2> $OUT/hpmc$i.err )
    f=$(find $OUT/hpmc$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "dsgd_hogwild" | tee -a $OUT/hogwild_pmc_summary.txt; else tail -5 $OUT/hpmc$i.err; fi
    rm -rf $OUT/hpmc$i $OUT/hpmc$i.err
  done ;;
esac
done
