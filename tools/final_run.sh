#!/bin/bash
# One GPU-box visit that produces everything committed under profiles/ for a round:
#   GPU parity tests, the default bench line, a rocprofv3 kernel trace of the same step, two PMC passes.
# usage (repo root, on the GPU box): bash tools/final_run.sh <tag>
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
{ nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; rocm-smi --showproductname 2>/dev/null | head -8; } > $OUT/env.txt 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -80 | tee $OUT/pytest_gpu.txt
echo "== bench (default flags)"
timeout 900 python bench.py 2> $OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err; cut -c1-400 $OUT/bench.json
echo "== rocprofv3 kernel trace"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep --no-parity-gate > $OUT/bench_prof.json 2> $OUT/bench_prof.err )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -8 "$f" | cut -c1-160
i=0
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "== pmc pass $i: $P"
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python $REPO/tools/pmc_step.py 8388608 3 > $OUT/pmc$i.out 2> $OUT/pmc$i.err )
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" "dsgd_" | grep -E "wseg|cdot|cgrad|reduce|apply" | tee -a $OUT/pmc_summary.txt
  # the raw per-dispatch file is large: keep only the summary
  rm -rf $OUT/pmc$i
done
