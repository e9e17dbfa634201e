OUT=$PWD/gpurun_out/r02f; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
DSGD_COLD8=3 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "streaming_layouts or full_size or ragged" 2>&1 | tail -3
timeout 300 python tools/plan_prof.py 2> $OUT/plan_prof.err | tee $OUT/plan_prof.json | python -c "
import json,sys
d=json.load(sys.stdin)
for p in d['plan']: print('plan', p['workers'], p['batch'], round(p['us_per_step'],2), {k:round(v) for k,v in p['cycles_per_step'].items()})
"
