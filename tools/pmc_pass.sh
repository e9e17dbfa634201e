#!/bin/bash
# rocprofv3 --pmc passes (one per argument) over tools/pmc_step.py; prints per-kernel averages.
# usage: bash tools/pmc_pass.sh <tag> "<counters>" ["<counters>" ...]
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for P in "$@"; do
  i=$((i+1))
  echo "== pmc pass $i: $P" | tee -a $OUT/pmc_summary.txt
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python $REPO/tools/pmc_step.py 8388608 3 > $OUT/pmc$i.out 2> $OUT/pmc$i.err )
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "dsgd_" | grep -E "fstep|wseg|cold|reduce_apply" | tee -a $OUT/pmc_summary.txt; else tail -3 $OUT/pmc$i.err; fi
  rm -rf $OUT/pmc$i
done
