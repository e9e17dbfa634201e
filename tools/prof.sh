#!/bin/bash
# rocprofv3 kernel-trace summary of a short bench run.  usage: bash tools/prof.sh <tag> [bench args...]
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/bench.py --gpus 1 --no-cpu-baseline "$@" > $OUT/bench_prof.json 2> $OUT/bench_prof.err
cd $REPO
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; head -20 "$f"; else echo "no kernel_stats"; tail -5 $OUT/bench_prof.err; find $OUT/prof | head; fi
cat $OUT/bench_prof.json | head -c 3000
