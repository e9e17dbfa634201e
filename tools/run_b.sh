OUT=gpurun_out/r02b; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -30 | tee $OUT/pytest_gpu.txt
timeout 300 python tools/plan_prof.py 2> $OUT/plan_prof.err | tee $OUT/plan_prof.json
timeout 600 python bench.py 2> $OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err; cut -c1-300 $OUT/bench.json
