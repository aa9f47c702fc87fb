#!/bin/bash
# Same-box A/B of several tagged library builds (gaussiansplat_amd/lib_TAG/): C3 frames and the isolated composite kernels, interleaved.
#   tools/ab_multi.sh lib lib_u3m5 lib_u4m4
set -e -o pipefail
for rep in 1 2; do
  for d in "$@"; do
    lib=$PWD/gaussiansplat_amd/$d/libgsplat_hip.so
    GSPLAT_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --config C3 --no-cpu-baseline --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$d', 'C3 ms/frame %.4f' % d['ms_per_step'], 'fwd %.3f bwd %.3f' % (s['composite_fwd'], s['composite_bwd']))"
  done
done
for rep in 1 2; do
  for d in "$@"; do
    echo "== isolated kernels $d"
    GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/$d/libgsplat_hip.so AB_TMIN=1e-5 AB_ROUNDS=4 timeout -k 10 120 python3 tools/abtest.py C3 30 30 2>&1 | grep -v amdgpu.ids
  done
done
