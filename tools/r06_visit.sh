#!/bin/bash
# One GPU-box visit of round 6 (legs of tools/r05_visit.sh / r04_visit.sh are available by name).
# usage (repo root, on the GPU box): bash tools/r06_visit.sh <tag> [testall] [bench] [benchprof] [pmc] [densepmc] [fstepab] [smoke] ...
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
for leg in "$@"; do
case $leg in
densepmc)
  # the dense variant on the VALU and on the matrix cores (DSGD_DENSE_MFMA=1): instruction mix and busy cycles per launch
  for M in 0 1; do
    echo "== DSGD_DENSE_MFMA=$M" | tee -a $OUT/dense_pmc_summary.txt
    for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
      ( cd /tmp && DSGD_DENSE_MFMA=$M timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/dpmc -o pmc -- python $REPO/tools/dense_check.py 300000 > $OUT/dpmc.out 2> $OUT/dpmc.err )
      f=$(find $OUT/dpmc -name "*counter_collection.csv" | head -1)
      if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "dsgd_dense_step" | tee -a $OUT/dense_pmc_summary.txt; else tail -3 $OUT/dpmc.err | tee -a $OUT/dense_pmc_summary.txt; fi
      rm -rf $OUT/dpmc
    done
    ( cd /tmp && DSGD_DENSE_MFMA=$M timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dst -o st -- python $REPO/tools/dense_check.py 300000 > /dev/null 2> $OUT/dst.err )
    f=$(find $OUT/dst -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "dense_step" "$f" | cut -c1-200 | tee -a $OUT/dense_pmc_summary.txt
    rm -rf $OUT/dst
  done ;;
fstepab)
  bash tools/r06_ab.sh 804414,100552 $TAG; cp gpurun_out/r06_ab_$TAG.txt $OUT/fstep_ab.txt; cut -c1-150 $OUT/fstep_ab.txt ;;
fitloop)
  timeout 600 python tools/fit_loop_prof.py 804414 6 > $OUT/fit_loop.txt 2>&1; timeout 300 python tools/fit_loop_prof.py 23149 10 >> $OUT/fit_loop.txt 2>&1; cut -c1-250 $OUT/fit_loop.txt ;;
moreprof)
  # rocprofv3 kernel statistics of (1) host.MasterSync.fit at N = 804,414, 3 x 100 with the lists drawn by the device, (2) whole-split steps at N = 804,414
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p1 -o t -- python $REPO/tools/fit_loop_prof.py 804414 3 > $OUT/fit_prof.out 2> $OUT/fit_prof.err )
  f=$(find $OUT/p1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/fit_kernel_stats.csv && grep -E "jr_|cs_step|cs_layout|eval|Name" "$f" | cut -c1-40,100-400 | head -12
  rm -rf $OUT/p1
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p2 -o t -- python $REPO/tools/fstep_prof.py 804414 1 200 > $OUT/step_prof.out 2> $OUT/step_prof.err )
  f=$(find $OUT/p2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/step804_kernel_stats.csv && grep -E "fstep_kernel|reduce_apply|Name" "$f" | cut -c1-40,300-500 | head -6
  rm -rf $OUT/p2
  for P in "FETCH_SIZE" "WRITE_SIZE"; do
    ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p3 -o pmc -- python $REPO/tools/pmc_step.py 804414 20 > $OUT/p3.out 2> $OUT/p3.err )
    f=$(find $OUT/p3 -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" "dsgd_" | grep -E "fstep_kernel|reduce_apply" | tee -a $OUT/step804_pmc.txt
    rm -rf $OUT/p3
  done ;;
*)
  bash tools/r05_visit.sh $TAG $leg ;;
esac
done
