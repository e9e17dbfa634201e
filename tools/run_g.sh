OUT=$PWD/gpurun_out/r02g; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
timeout 300 python tools/plan_prof.py 2> $OUT/plan_prof.err | tee $OUT/plan_prof.json | python -c "
import json,sys
d=json.load(sys.stdin)
for p in d['plan']: print('plan', p['workers'], p['batch'], round(p['us_per_step'],2), {k:round(v) for k,v in p['cycles_per_step'].items()})
"
timeout 600 python bench.py --no-cpu-baseline 2> $OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err; python -c "
import json
d=json.load(open('$OUT/bench.json'))
print(round(d['value']/1e9,3), d['ms_per_step']); print(json.dumps(d['sweep'])); print(json.dumps(d['hogwild'])); print(json.dumps(d['dense_logistic']))
"
