#!/bin/bash
# Same-box A/B of two libraries on C3 and C5 frames and on the isolated composite kernels (interleaved).  tools/ab_two.sh LIB_A LIB_B [configs]
A=$PWD/$1; B=$PWD/$2; CFGS=${3:-"C3 C5"}
for rep in 1 2 3; do
  for lib in $A $B; do
    for c in $CFGS; do
      GSPLAT_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$(basename $(dirname $lib))', '$c', 'ms/frame %.4f' % d['ms_per_step'], 'fwd %.4f bwd %.4f tile_sort %.4f' % (s['composite_fwd'], s['composite_bwd'], s['tile_sort']))"
    done
  done
done
for lib in $A $B $A $B; do
  echo "== isolated kernels $(basename $(dirname $lib))"
  GSPLAT_HIP_LIB=$lib AB_TMIN=1e-5 AB_ROUNDS=4 timeout -k 10 120 python3 tools/abtest.py C3 30 30 2>&1 | grep -v amdgpu.ids
done
