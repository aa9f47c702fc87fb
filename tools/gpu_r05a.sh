#!/bin/bash
# round 5, first GPU call: occupancy curve of the composite kernels, forced parts, baseline bench lines of this box
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/occupancy_curve.py C3 gpurun_out/r05a_occupancy_curve_C3.json > gpurun_out/r05a_occ.log 2>&1; echo "occ rc=$?"
timeout -k 10 200 python3 tools/forced_parts.py C3 > gpurun_out/r05a_forced_parts_C3.log 2>&1; echo "parts rc=$?"
mv gpurun_out/forced_parts_C3.json gpurun_out/r05a_forced_parts_C3.json
for c in C3 C1 C2; do
  steps=20; [ $c != C3 ] && steps=60
  timeout -k 10 200 python3 bench.py --config $c --steps $steps --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor > gpurun_out/r05a_bench_$c.json 2> gpurun_out/r05a_bench_$c.err; echo "bench $c rc=$?"
done
grep -v amdgpu gpurun_out/r05a_occ.log | cut -c1-400
tail -30 gpurun_out/r05a_forced_parts_C3.log
python3 - <<'PY'
import json
for c in ("C3","C1","C2"):
    try:
        d=json.loads(open(f"gpurun_out/r05a_bench_{c}.json").read().strip().splitlines()[-1]); print(c, d["ms_per_step"], {k: round(v,4) for k,v in d["stage_ms"].items() if v>0})
    except Exception as e: print(c,"ERR",e)
PY
