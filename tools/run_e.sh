OUT=$PWD/gpurun_out/r02e; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
timeout 300 python tools/plan_prof.py 2> $OUT/plan_prof.err | tee $OUT/plan_prof.json | python -c "
import json,sys
d=json.load(sys.stdin)
for p in d['plan']: print('plan', p['workers'], p['batch'], round(p['us_per_step'],2), {k:round(v) for k,v in p['cycles_per_step'].items()})
for h in d['hogwild']: print('hog', h['workers'], round(h['examples_per_s']/1e6,1), round(h['us_per_iteration_per_worker'],1))
"
for v in 1 0; do
DSGD_COLD8=$v timeout 600 python bench.py --no-cpu-baseline --no-sweep 2> $OUT/bench_cold$v.err > $OUT/bench_cold$v.json; python -c "
import json
d=json.load(open('$OUT/bench_cold$v.json'))
r=d['roofline']
print('cold8=$v', round(d['value']/1e9,3), 'G ex/s', round(d['ms_per_step'],4), 'ms; main', round(r['kernel_ms_avg'],4), 'frac', round(r['frac'],3), 'step frac', round(r['step']['frac'],3), {k:(round(v['ms_avg'],4), round(v['achieved'])) for k,v in r['other_kernels'].items()})
"
done
