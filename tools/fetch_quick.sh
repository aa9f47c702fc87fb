#!/bin/bash
# FETCH_SIZE of the composite kernels inside real frames (bench.py C3):  tools/fetch_quick.sh [libdir]
set -e -o pipefail
export TMPDIR=/tmp
[ -n "$1" ] && export GSPLAT_HIP_LIB=$PWD/$1/libgsplat_hip.so
D=$PWD/gpurun_out/fetchq_$(basename ${1:-lib}); mkdir -p $D
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D -- python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-train-iteration --no-c4-anchor --no-literal > $D/run.log 2>&1
python3 - "$D" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "composite_" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"][:44]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = v[-16:]
    print(sys.argv[1].split("_")[-1], k, "FETCH_SIZE x2 per launch: %.0f MB" % (2 * 1024 * sum(v) / len(v) / 1e6))
PY
find $D -name "*counter_collection.csv" -delete
