#!/bin/bash
# Same-box A/B of two whole libraries (GSPLAT_HIP_LIB): frames at C1 / C2 / C3 and the isolated composite kernels, interleaved.
#   tools/ab_libs.sh gaussiansplat_amd/lib_old/libgsplat_hip.so gaussiansplat_amd/lib/libgsplat_hip.so
set -e -o pipefail
A=$PWD/$1; B=$PWD/$2
for rep in 1 2; do
  for lib in $A $B; do
    for c in C3 C2 C1; do
      GSPLAT_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --config $c --no-cpu-baseline --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$(basename $(dirname $lib))', '$c', 'ms/frame %.4f' % d['ms_per_step'], 'fwd %.3f bwd %.3f' % (s['composite_fwd'], s['composite_bwd']))"
    done
  done
done
for lib in $A $B $A $B; do
  echo "== isolated kernels $(basename $(dirname $lib))"
  GSPLAT_HIP_LIB=$lib AB_TMIN=1e-5 AB_ROUNDS=4 timeout -k 10 120 python3 tools/abtest.py C3 30 30 2>&1 | grep -v amdgpu.ids
done
