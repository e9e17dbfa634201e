#!/bin/bash
# One GPU visit of round 5: tests first (the new column-slice work, then the whole GPU suite), the probe, a bench line.
# usage: tools/r05_visit.sh <tag> [quick]
tag=${1:-v}
mode=${2:-full}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${tag}_build.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_cs_device.py tests/test_gpu_cs.py -q -m gpu --tb=short > gpurun_out/${tag}_pytest_cs.txt 2>&1
tail -5 gpurun_out/${tag}_pytest_cs.txt
timeout 600 python tools/r05_probe.py > gpurun_out/${tag}_probe.json 2> gpurun_out/${tag}_probe.err
tail -3 gpurun_out/${tag}_probe.err
if [ "$mode" = "full" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/${tag}_pytest_gpu.txt 2>&1
  tail -8 gpurun_out/${tag}_pytest_gpu.txt
fi
