#!/bin/bash
# A/B on ONE box: whole-split steps through the round-5 library (tools/ab/libdsgd_hip_r05.so, built from c64ce6f) and the tree's
sizes=${1:-804414}
out=gpurun_out/r06_ab_${2:-a}.txt
: > $out
for rep in 1 2; do
  echo "== r05 library" >> $out
  DSGD_LIB_PATH=$PWD/tools/ab/libdsgd_hip_r05.so timeout 600 python tools/fstep_prof.py $sizes 1 300 >> $out 2>&1
  echo "== this tree" >> $out
  timeout 600 python tools/fstep_prof.py $sizes 1 300 >> $out 2>&1
done
