#!/bin/bash
# Where do the forward's 483 MB per launch (against 369 MB algorithmic) come from?  One --pmc pass per variant (GPU box):
#   production order (8 x 8-tile groups per XCD) / tile order (neighbours on eight XCDs) / a build that gathers nothing ahead of the early-out.
export TMPDIR=/tmp
TAG=${1:-r05}
[ -f gaussiansplat_amd/lib_nopf/libgsplat_hip.so ] || { echo "build the variant first: python -m gaussiansplat_amd.build --tag nopf -DGS_FWD_NO_PREFETCH=1"; exit 1; }
O=$PWD/gpurun_out/fwdsplit_$TAG; rm -rf $O; mkdir -p $O
FWD_SPLIT_HOST_STATS=1 python3 tools/fwd_traffic_split.py 30 4 2>/dev/null | tail -1 > $O/host_stats.json
for v in "30 lib" "10 lib" "30 lib_nopf"; do
  set -- $v
  export GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/$2/libgsplat_hip.so
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --output-format csv -d $O/${2}_v$1_$n -- python3 tools/fwd_traffic_split.py $1 20 > $O/${2}_v$1_$n.log 2>&1
  done
  python3 tools/pmc_summary.py $O/${2}_v$1.json $O/${2}_v$1_FETCH_SIZE $O/${2}_v$1_TCC_HIT_sum --match composite_fwd > /dev/null 2>&1
done
unset GSPLAT_HIP_LIB
python3 - "$O" "$TAG" <<'PY'
import json, sys, glob, os
O, tag = sys.argv[1], sys.argv[2]
out = {"host": json.load(open(O + "/host_stats.json"))}
for f in sorted(glob.glob(O + "/lib*_v*.json")):
    d = json.load(open(f))
    for k, v in d["kernels"].items():
        out[os.path.basename(f)[:-5]] = {kk: v.get(kk) for kk in ("hbm_bytes_fetch_x2", "hbm_bytes_fetch", "FETCH_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "mean_ns_under_pmc", "dispatches")}
json.dump(out, open("gpurun_out/%s_fwd_traffic_split.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*.csv" -size +2M -delete
