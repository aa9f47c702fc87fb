#!/bin/bash
# Same-box A/B against an earlier commit's kernels:  tools/build_old_lib.sh COMMIT [TAG=old]  ->  gaussiansplat_amd/lib_TAG/libgsplat_hip.so
# built from that commit's csrc/ + include/ with the flags of gaussiansplat_amd/build.py; ABI functions added since (the ctypes binding
# refuses a library without them) get a stub.  Then: GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/lib_TAG/libgsplat_hip.so python3 bench.py ...
set -e -o pipefail
C=$1; TAG=${2:-old}
W=$(mktemp -d)
git archive "$C" gaussiansplat_amd/csrc include | tar -x -C "$W"
OUT=$PWD/gaussiansplat_amd/lib_$TAG; mkdir -p "$OUT"
grep -q gs_get_bin_path "$W/include/gsplat.h" || echo 'extern "C" int gs_get_bin_path(gs_ctx *c) { return c ? 0 : -1; }' >> "$W/gaussiansplat_amd/csrc/gs_api_debug.hip"
objs=()
for s in "$W"/gaussiansplat_amd/csrc/*.hip; do
  b=$(basename "$s" .hip); extra=""
  case $b in gs_preprocess|gs_preprocess2d) extra="-ffp-contract=off";; gs_composite|gs_loss) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function $extra -c "$s" -o "$OUT/$b.o" &
  objs+=("$OUT/$b.o")
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libgsplat_hip.so" "${objs[@]}" -ldl
rm -rf "$W"; echo "$OUT/libgsplat_hip.so"
