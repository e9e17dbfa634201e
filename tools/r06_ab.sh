#!/bin/bash
# A/B on ONE box: whole-split steps through the round-5 library and the tree's.  The round-5 library is not in the history
# (built artefacts are git-ignored); build it where hipcc is, it travels with the snapshot:
#   mkdir -p /tmp/r05src tools/ab && git archive c64ce6f distributed-sgd_amd/csrc include | tar -x -C /tmp/r05src && (cd /tmp/r05src &&
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -w -Iinclude distributed-sgd_amd/csrc/dsgd_hip.hip \
#         -o $OLDPWD/tools/ab/libdsgd_hip_r05.so -Wl,-rpath,/opt/rocm/lib -ldl -lpthread -L/opt/rocm/lib -lamdhip64)
sizes=${1:-804414}
out=gpurun_out/r06_ab_${2:-a}.txt
: > $out
for rep in 1 2; do
  echo "== r05 library" >> $out
  DSGD_LIB_PATH=$PWD/tools/ab/libdsgd_hip_r05.so timeout 600 python tools/fstep_prof.py $sizes 1 300 >> $out 2>&1
  echo "== this tree" >> $out
  timeout 600 python tools/fstep_prof.py $sizes 1 300 >> $out 2>&1
done
