#!/bin/bash
# GPU box: the suite, then the C3 / C5 / C2 bench lines and the isolated composite kernel times.  usage: bash tools/gpu_quick.sh TAG [pytest args]
tag=${1:-q}; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x "$@" > gpurun_out/${tag}_tests.log 2>&1; echo rc=$? >> gpurun_out/${tag}_tests.log; tail -4 gpurun_out/${tag}_tests.log
grep -q "rc=0" gpurun_out/${tag}_tests.log || { grep -n "Error\|assert" gpurun_out/${tag}_tests.log | head -20; }
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor > gpurun_out/${tag}_c3.json 2>gpurun_out/${tag}_b.err
python3 bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline --no-literal > gpurun_out/${tag}_c5.json 2>>gpurun_out/${tag}_b.err
python3 bench.py --config C2 --steps 40 --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration > gpurun_out/${tag}_c2.json 2>>gpurun_out/${tag}_b.err
AB_ROUNDS=4 AB_TMIN=1e-5 python3 tools/abtest.py C3 30 30 > gpurun_out/${tag}_ab_c3.log 2>&1
python3 - "$tag" <<'PY'
import glob, json, sys
tag = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/%s_c[235].json" % tag)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(f, "ms", round(d["ms_per_step"], 4), "unsettled", d.get("ms_per_step_unsettled"), "noslots", (d.get("no_view_slot_history") or {}).get("ms_per_step"), "capped", c["lists_capped"], {k: v for k, v in d["stage_ms"].items() if v > 0},
              "dom", round(d["roofline"]["avg_ms"], 4))
    except Exception as e:
        print(f, "ERR", e)
PY
cat gpurun_out/${tag}_ab_c3.log | tail -4
