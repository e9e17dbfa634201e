OUT=$PWD/gpurun_out/r02i; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline 2> $OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err; python -c "
import json
d=json.load(open('$OUT/bench.json'))
r=d['roofline']
print(round(d['value']/1e9,3), d['ms_per_step'], r['kernel_ms_avg'], r['frac'], r['step']['frac']); print(json.dumps(d['dense_logistic']['batches'][0])); print(json.dumps(d['dense_logistic'].get('mfma_variant')))
"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep --no-parity-gate > $OUT/bench_prof.json 2> $OUT/bench_prof.err )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "reduce_apply|wseg_kernel<true|cdot8|cgrad" "$f" | cut -c1-40,200-330
rm -rf $OUT/prof
