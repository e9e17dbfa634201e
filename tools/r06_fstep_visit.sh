#!/bin/bash
# one GPU visit for the row-chunk kernel: parity, wall time per whole-split step at the reference's sizes, phase cycles, per-workgroup times
tag=${1:-a}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fstep.py -x -q 2>&1 | tail -5 > gpurun_out/r06_fstep_${tag}_pytest.txt
timeout 600 python tools/fstep_prof.py 100552,804414,2000000 1 200 > gpurun_out/r06_fstep_${tag}_plain.txt 2>&1
DSGD_PLAN_PROF=1 timeout 600 python tools/fstep_prof.py 804414 1 100 > gpurun_out/r06_fstep_${tag}_phases.txt 2>&1
rm -f gpurun_out/wg804_${tag}.txt
DSGD_PLAN_PROF=1 DSGD_FSTEP_DUMP=gpurun_out/wg804_${tag}.txt timeout 600 python tools/fstep_wgdump.py 804414 > /dev/null 2>&1
python tools/fstep_wganalyse.py gpurun_out/wg804_${tag}.txt > gpurun_out/r06_fstep_${tag}_wg.txt 2>&1
