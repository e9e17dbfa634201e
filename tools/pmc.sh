#!/bin/bash
# rocprofv3 PMC passes (each in its own run, --kernel-trace only) of a short bench run.
# usage: bash tools/pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- [bench args]
TAG=$1; shift
PASSES=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do PASSES+=("$1"); shift; done
shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  cd /tmp
  timeout 900 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python $REPO/bench.py --gpus 1 --no-cpu-baseline --no-sweep "$@" > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  cd $REPO
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "seg_kernel" in k or "stream_kernel" in k or "cold_scatter" in k:
        print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
  else echo "no counter file"; tail -3 $OUT/pmc$i.err; fi
done
