#!/bin/bash
# Diagnostic builds of the composite backward with parts removed (GS_ABL bit mask in gs_composite.hip), timed on the GPU box:
#   tools/ablate.sh build      (here: cross-compiles tools/abl/libgsplat_<tag>.so; needs `python -m gaussiansplat_amd.build --experiments` first)
#   tools/ablate.sh run        (GPU box: times body 3, plain scheduling, of each build at t_min = 0 with tools/abtest.py)
set -e
cd "$(dirname "$0")/.."
D=tools/abl
mkdir -p $D
# tag:GS_ABL:extra flags
SPECS="${SPECS:-full:0: nored:1: noatomic:2: nored_notrans:9: nored_nobox:17: nored_notrans_nobox:25: nored_nolds:33: nored_all:57: nored_noslp:1:-fno-slp-vectorize full_noslp:0:-fno-slp-vectorize nored_w8:1:-DGS_BWD3_MINW=8 full_w8:0:-DGS_BWD3_MINW=8}"
if [ "$1" = build ]; then
  for spec in $SPECS; do
    IFS=: read tag abl flags <<< "$spec"
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -DGS_EXPERIMENTS -DGS_ABL=$abl $flags -c gaussiansplat_amd/csrc/gs_composite.hip -o $D/gs_composite_$tag.o &
  done
  wait
  for spec in $SPECS; do
    IFS=: read tag abl flags <<< "$spec"
    objs=$(ls gaussiansplat_amd/lib_exp/*.o | grep -v gs_composite.o)     # GS_EXPERIMENTS objects: GS_DEBUG_EXTRA_LDS, variants
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $D/libgsplat_$tag.so $objs $D/gs_composite_$tag.o -ldl
  done
elif [ "$1" = occ ]; then
  # occupancy sweep of the shipped build and the no-reduction build: extra dynamic LDS caps the waves per CU
  for tag in full nored; do
    for extra in 0 1900 3900 7200 13900 33900; do
      echo "== $tag extra LDS $extra B  (waves/CU <= $(( 163840 / (6100 + extra) )))"
      GS_DEBUG_EXTRA_LDS=$extra GSPLAT_HIP_LIB=$PWD/$D/libgsplat_$tag.so AB_TMIN=${AB_TMIN:-0} AB_ROUNDS=3 python3 tools/abtest.py C3 ${AB_F:-10} ${AB_B:-13} 2>&1 | grep -v amdgpu.ids
    done
  done
else
  for spec in $SPECS; do
    IFS=: read tag abl flags <<< "$spec"
    echo "== $tag (GS_ABL=$abl $flags)"
    GSPLAT_HIP_LIB=$PWD/$D/libgsplat_$tag.so AB_TMIN=${AB_TMIN:-0} AB_ROUNDS=3 python3 tools/abtest.py C3 ${AB_F:-10} ${AB_B:-13} 2>&1 | grep -v amdgpu.ids
  done
fi
