#!/bin/bash
# GPU box: the whole -m gpu suite, then same-box A/B of the default build against `--list-cap 1` at C3 and C5 (round 4).
# usage (from the repo root, through gpurun): bash tools/gpu_check.sh [tag]
tag=${1:-chk}
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --durations=8 > gpurun_out/${tag}_tests.log 2>&1; echo rc=$? >> gpurun_out/${tag}_tests.log
tail -15 gpurun_out/${tag}_tests.log
for rep in 1 2; do
for c in "" "--list-cap 1"; do
  k=$(echo "$c" | tr -d " -")
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor $c > gpurun_out/${tag}_c3${k}_$rep.json 2>gpurun_out/${tag}_b.err
  python bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline --no-literal $c > gpurun_out/${tag}_c5${k}_$rep.json 2>>gpurun_out/${tag}_b.err
done
done
python - "$tag" <<'PY'
import glob, json, sys
for f in sorted(glob.glob("gpurun_out/%s_c*.json" % sys.argv[1])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d["config"]
        print(f, "ms", round(d["ms_per_step"], 4), "capped", c["lists_capped"], "listed", c["listed_entries"], "of", c["instances"], "ext", c["list_segments_appended_by_waves"],
              {k: v for k, v in d["stage_ms"].items() if v > 0})
    except Exception as e:
        print(f, "ERR", e)
PY
