#!/bin/bash
# whole-split steps with and without the measured re-cut of the row chunks, on ONE box (tools/fstep_prof.py: 20 untimed + N timed steps)
out=gpurun_out/r06_rebalance_${1:-a}.txt
: > $out
for rep in 1 2; do
  for m in 0 1; do
    echo "== DSGD_FSTEP_REBALANCE=$m" >> $out
    DSGD_FSTEP_REBALANCE=$m timeout 600 python tools/fstep_prof.py 804414,2000000,8388608 1 300 200 >> $out 2>&1
  done
done
