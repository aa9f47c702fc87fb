#!/bin/bash
tag=${1:-r04c}
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_caps.py tests/test_gpu_multiview.py tests/test_gpu_slabs.py tests/test_gpu_schedule.py tests/test_gpu_literal.py -q -x > gpurun_out/${tag}_tests.log 2>&1; echo rc=$? >> gpurun_out/${tag}_tests.log; tail -4 gpurun_out/${tag}_tests.log
python3 tools/tile_tail.py C3 gpurun_out/${tag}_tile_tail_C3.json > gpurun_out/${tag}_tail_C3.log 2>&1
python3 tools/tile_tail.py C5 gpurun_out/${tag}_tile_tail_C5.json > gpurun_out/${tag}_tail_C5.log 2>&1
python3 tools/clock_ramp.py gpurun_out/${tag}_clock_ramp.json > gpurun_out/${tag}_clock_ramp.log 2>&1
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor > gpurun_out/${tag}_c3.json 2>gpurun_out/${tag}_b.err
for c in "" "--list-cap 1"; do k=$(echo "$c" | tr -d " -"); python3 bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline --no-literal $c > gpurun_out/${tag}_c5${k}.json 2>>gpurun_out/${tag}_b.err; done
python3 - "$tag" <<'PY'
import glob, json, sys
tag = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/%s_tile_tail_*.json" % tag)):
    d = json.load(open(f))
    print(f, json.dumps(d["fwd_v30"]["frozen_pixels"]))
for f in sorted(glob.glob("gpurun_out/%s_c[35]*.json" % tag)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(f, "ms", round(d["ms_per_step"], 4), "unsettled", d.get("ms_per_step_unsettled"), "noslots", d.get("no_view_slot_history"), "capped", c["lists_capped"], c["listed_entries"], {k: v for k, v in d["stage_ms"].items() if v > 0})
    except Exception as e:
        print(f, "ERR", e)
PY
cat gpurun_out/${tag}_clock_ramp.log | cut -c1-1500
