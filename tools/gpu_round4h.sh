#!/bin/bash
tag=${1:-r04h}
export TMPDIR=/tmp
python -m pytest tests -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; echo rc=$? >> gpurun_out/${tag}_tests.log; tail -3 gpurun_out/${tag}_tests.log
grep -q "rc=0" gpurun_out/${tag}_tests.log || { grep -n "Error\|assert\|FAILED" gpurun_out/${tag}_tests.log | head -20; }
for rep in 1 2 3; do for f in 0 8; do python3 bench.py --config C3 --steps 20 --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor --debug-flags $f 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 debug_flags $f', round(d['ms_per_step'],4), {k:v for k,v in d['stage_ms'].items() if v>0}, d['config']['coarse_instances'])"; done; done
for c in C1 C2 C4; do python3 bench.py --config $c --steps 40 --warmup 6 --no-cpu-baseline --no-literal --no-train-iteration 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],4), round(d['value'],1), {k:v for k,v in d['stage_ms'].items() if v>0})"; done
