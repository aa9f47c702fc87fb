#!/bin/bash
# per-kernel time of one bench command (GPU box):  tools/kstats.sh TAG [bench args...]  -> gpurun_out/kstats_TAG.csv
set -e -o pipefail
TAG=$1; shift
OUT=$PWD/gpurun_out/kstats_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $PWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-literal --no-train-iteration "$@" > "$OUT/bench.log" 2>&1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$PWD/gpurun_out/kstats_$TAG.csv" \;
find "$OUT" -name "*kernel_trace.csv" -delete
grep '^{"metric"' "$OUT/bench.log" > "$PWD/gpurun_out/kstats_$TAG.json" || true
