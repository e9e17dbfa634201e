OUT=$PWD/gpurun_out/r02h; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
timeout 300 python tools/plan_prof.py 2> $OUT/plan_prof.err | tee $OUT/plan_prof.json | python -c "
import json,sys
d=json.load(sys.stdin)
for p in d['plan']: print('plan', p['workers'], p['batch'], round(p['us_per_step'],2), {k:round(v) for k,v in p['cycles_per_step'].items()})
for h in d['hogwild']: print('hog', h['workers'], round(h['examples_per_s']/1e6,1), round(h['us_per_iteration_per_worker'],1))
"
timeout 300 python tools/hog_prof.py 8388608 256 60000 2> $OUT/hog.err | tee $OUT/hog_8m.json
