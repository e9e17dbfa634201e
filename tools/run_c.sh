OUT=$PWD/gpurun_out/r02c; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline 2> $OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err; cut -c1-300 $OUT/bench.json
( cd /tmp && rocprofv3 -L > $OUT/counters.txt 2>&1 ); grep -iE "ATOMIC|UTCL|TLB" $OUT/counters.txt | head -60 > $OUT/counters_sel.txt; wc -l $OUT/counters.txt
timeout 300 python tools/hog_prof.py 8388608 256 40000 > $OUT/hog_8m.json 2> $OUT/hog_8m.err; cat $OUT/hog_8m.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hogtrace -o hog -- python $REPO/tools/hog_prof.py 8388608 256 40000 > $OUT/hog_trace.json 2> $OUT/hog_trace.err )
f=$(find $OUT/hogtrace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|hogwild" "$f" | cut -c1-200
i=0
for P in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TA_BUSY_avr"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/hpmc$i -o pmc -- python $REPO/tools/hog_prof.py 8388608 256 40000 > $OUT/hpmc$i.out 2> $OUT/hpmc$i.err )
  f=$(find $OUT/hpmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "dsgd_hogwild" | tee -a $OUT/hog_pmc_summary.txt; else tail -5 $OUT/hpmc$i.err; fi
  rm -rf $OUT/hpmc$i
done
