#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the composite kernels under the launch orders of tools/xcd_order.py (experiments build, GPU box)
set -e -o pipefail
export TMPDIR=/tmp GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/lib_exp/libgsplat_hip.so
for o in mod8 super8 super4; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    D=$PWD/gpurun_out/xcd_$o/$ctr; mkdir -p $D
    rocprofv3 --pmc $ctr --output-format csv -d $D -- python3 tools/xcd_order.py C3 single:$o > $D/run.log 2>&1
  done
  python3 - "$o" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/xcd_{o}/{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "composite_" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[-8:]                                                   # the timed launches of the custom order (the last ones)
        print(o, ctr, k, "KB per launch (last 8): %.0f" % (sum(v) / len(v)), "x 64 B units" if ctr == "FETCH_SIZE" else "")
PY
  find gpurun_out/xcd_$o -name "*counter_collection.csv" -delete
done
