#!/bin/bash
# L2 (TCC) request counters of the lock-free engine's persistent kernel: separate --pmc passes over tools/hog_prof.py
# (two launches of 30,000 updates each at 256 workers x 100).  usage (repo root, GPU box): bash tools/r06_hog_pmc.sh <tag> [workers]
TAG=$1; W=${2:-256}
OUT=$PWD/gpurun_out/$TAG; REPO=$PWD
mkdir -p $OUT; export TMPDIR=/tmp
: > $OUT/hog_pmc_summary.txt
for P in "TCC_REQ_sum TCC_ATOMIC_sum TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_ATOMIC_WITH_RET_REQ_sum TCC_READ_REQ_sum TCC_READ_REQ_LATENCY_sum" "TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_NC_REQ_sum TCC_CC_REQ_sum"; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/hp -o pmc -- python $REPO/tools/hog_prof.py 2097152 $W 30000 > $OUT/hp.out 2> $OUT/hp.err )
  f=$(find $OUT/hp -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "dsgd_hogwild" | tee -a $OUT/hog_pmc_summary.txt; else echo "no counters for: $P"; tail -3 $OUT/hp.err; fi | cut -c1-400
  grep -o '"updates": [0-9]*, "ms": [0-9.]*' $OUT/hp.out | tee -a $OUT/hog_pmc_summary.txt
  rm -rf $OUT/hp
done
