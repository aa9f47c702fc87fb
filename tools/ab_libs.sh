#!/bin/bash
# Same-box A/B of several library builds: C3 frames (interleaved, 3 rounds) and the isolated composite kernels.
#   tools/ab_libs.sh lib lib_x lib_y ...      (directory names under gaussiansplat_amd/)
CFGS=${AB_CFGS:-C3}
for rep in 1 2 3; do
  for d in "$@"; do
    lib=$PWD/gaussiansplat_amd/$d/libgsplat_hip.so
    for c in $CFGS; do
      GSPLAT_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%-10s' % '$d', '$c', 'ms/frame %.4f' % d['ms_per_step'], 'fwd %.4f bwd %.4f lists %.4f' % (s['composite_fwd'], s['composite_bwd'], s['tile_sort']))"
    done
  done
done
for rep in 1 2; do
  for d in "$@"; do
    echo "== isolated kernels $d"
    GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/$d/libgsplat_hip.so AB_TMIN=1e-5 AB_ROUNDS=4 timeout -k 10 120 python3 tools/abtest.py C3 30 30 2>&1 | grep -v amdgpu.ids
  done
done
