#!/bin/bash
# Same-box A/B of whole library builds (composite variants need a recompile, not a runtime flag):
#   tools/ab_libs.sh build TAG FILE.hip [extra hipcc flags]   here: tools/abl/libgsplat_TAG.so with csrc/<unit> replaced by FILE.hip
#   tools/ab_libs.sh run "TAG1 TAG2 ..." [C3] [fwd variants] [bwd variants]     GPU box: tools/abtest.py per build, twice, interleaved
# The unit replaced is the one FILE's basename starts with (gs_composite*.hip -> gs_composite.o, gs_bin3*.hip -> gs_bin3.o ...).
# tools/abl/*.so travel to the GPU box with the snapshot (git-ignored, not gpurun-ignored).
set -e
cd "$(dirname "$0")/.."
D=tools/abl; mkdir -p $D
if [ "$1" = build ]; then
  tag=$2; src=$3; shift 3
  unit=$(basename "$src" | sed -E 's/^(gs_[a-z0-9]+(_bwd)?).*/\1/')
  python3 -m gaussiansplat_amd.build > /dev/null
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Igaussiansplat_amd/csrc "$@" -c "$src" -o $D/${unit}_$tag.o
  objs=$(ls gaussiansplat_amd/lib/*.o | grep -v "/${unit}.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $D/libgsplat_$tag.so $objs $D/${unit}_$tag.o -ldl
  echo "$D/libgsplat_$tag.so ($unit from $src)"
elif [ "$1" = run ]; then
  tags=$2; cfg=${3:-C3}; vf=${4:-10}; vb=${5:-30}
  for rep in 1 2; do for t in $tags; do
    echo "== $t"
    lib=$PWD/$D/libgsplat_$t.so; [ "$t" = head ] && lib=$PWD/gaussiansplat_amd/lib/libgsplat_hip.so
    GSPLAT_HIP_LIB=$lib AB_TMIN=${AB_TMIN:-1e-5} AB_ROUNDS=${AB_ROUNDS:-8} timeout -k 10 300 python3 tools/abtest.py $cfg $vf $vb 2>&1 | grep -v amdgpu.ids
  done; done
else
  sed -n 2,8p "$0"
fi
