#!/bin/bash
# Four instead of five waves per SIMD for the composite kernels (forced through extra dynamic LDS): 8160 tiles are 1.6 per slot
# at 5 waves (half the slots run two tiles, half one: ragged end) but 1.99 per slot at 4.  Same box, interleaved.
set -e -o pipefail
export GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/lib_exp/libgsplat_hip.so AB_TMIN=1e-5 AB_ROUNDS=4
for rep in 1 2; do
  for extra in 0 3872 6912; do
    echo "== extra LDS $extra (rep $rep)"
    GS_DEBUG_EXTRA_LDS=$extra timeout -k 10 120 python3 tools/abtest.py C3 30 30 2>&1 | grep -v amdgpu.ids
  done
done
