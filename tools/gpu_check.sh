#!/bin/bash
# One GPU-box visit: environment facts, GPU parity tests, microbenchmarks, bench, rocprofv3 summary.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== host"; nproc; lscpu | grep -E "Model name|Socket|Thread|Core" ; free -g | head -2
  echo "== gpu"; rocm-smi --showproductname 2>/dev/null | head -12
} > $OUT/env.txt 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
echo "== microbench"
timeout 600 ./tools/microbench 1200000 2>&1 | tee $OUT/microbench.txt
echo "== bench"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 2> $OUT/bench.err | tee $OUT/bench.json
tail -5 $OUT/bench.err
echo "== rocprofv3 kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-sweep > $OLDPWD/$OUT/bench_prof.json 2> $OLDPWD/$OUT/bench_prof.err )
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
cat $OUT/env.txt
