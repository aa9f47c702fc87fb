#!/bin/bash
# GPU box, round 4: suite, per-kernel stats with and without list caps (C3, C5), frozen-pixel clocks, settle trace.
tag=${1:-r04b}
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x > gpurun_out/${tag}_tests.log 2>&1; echo rc=$? >> gpurun_out/${tag}_tests.log; tail -4 gpurun_out/${tag}_tests.log
bash tools/kstats.sh ${tag}_C3 --no-c4-anchor && bash tools/kstats.sh ${tag}_C3_nocap --no-c4-anchor --list-cap 1
bash tools/kstats.sh ${tag}_C5 --config C5 && bash tools/kstats.sh ${tag}_C5_nocap --config C5 --list-cap 1
python3 tools/tile_tail.py C3 gpurun_out/${tag}_tile_tail_C3.json > gpurun_out/${tag}_tail_C3.log 2>&1
python3 tools/tile_tail.py C5 gpurun_out/${tag}_tile_tail_C5.json > gpurun_out/${tag}_tail_C5.log 2>&1
python3 tools/tile_tail.py C2 gpurun_out/${tag}_tile_tail_C2.json > gpurun_out/${tag}_tail_C2.log 2>&1
S=$PWD/gpurun_out/${tag}_settle; mkdir -p $S
rocprofv3 --kernel-trace --output-format csv -d $S -o settle -- python3 tools/settle_trace.py 48 default > gpurun_out/${tag}_settle_default.log 2>&1
f=$(find $S -name "*kernel_trace.csv" | head -1); python3 tools/settle_analyse.py "$f" gpurun_out/${tag}_settle_frames.json; find $S -name "*.csv" -delete
GS_SETTLE_STAGES=1 python3 tools/settle_trace.py 48 default > gpurun_out/${tag}_settle_plain.log 2>&1
python3 tools/settle_trace.py 48 noslots > gpurun_out/${tag}_settle_noslots.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/kstats_%s_*.csv" % tag)):
    print(f)
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.2: print("   %-70s calls %4s avg_us %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
for f in sorted(glob.glob("gpurun_out/%s_tile_tail_*.json" % tag)):
    d = json.load(open(f))
    print(f, {k: d[k]["frozen_pixels"] for k in d if isinstance(d[k], dict) and "frozen_pixels" in d[k]})
PY
tail -2 gpurun_out/${tag}_settle_plain.log | cut -c1-600; tail -1 gpurun_out/${tag}_settle_noslots.log | cut -c1-400
