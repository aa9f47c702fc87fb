#!/bin/bash
tag=${1:-r04g}
export TMPDIR=/tmp
python -m pytest tests -q -m gpu -x > gpurun_out/${tag}_tests.log 2>&1; echo rc=$? >> gpurun_out/${tag}_tests.log; tail -3 gpurun_out/${tag}_tests.log
grep -q "rc=0" gpurun_out/${tag}_tests.log || { grep -n "Error\|assert\|FAILED" gpurun_out/${tag}_tests.log | head -20; }
bash tools/ab_two.sh gaussiansplat_amd/lib_ref/libgsplat_hip.so gaussiansplat_amd/lib/libgsplat_hip.so > gpurun_out/${tag}_ab_bwdpack.log 2>&1; cat gpurun_out/${tag}_ab_bwdpack.log
for rep in 1 2; do for f in 0 16; do python3 bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline --no-literal --debug-flags $f 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5 debug_flags $f', round(d['ms_per_step'],4), {k:v for k,v in d['stage_ms'].items() if v>0}, d['config']['coarse_instances'])"; done; done
./tools/graph_probe
