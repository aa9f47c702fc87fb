#!/bin/bash
# Same-box A/B of several libraries on whole bench lines.  tools/ab_many.sh "C4 C2" lib_a/libgsplat_hip.so lib_b/libgsplat_hip.so ...
CFGS=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    for c in $CFGS; do
      steps=20; [ $c = C5 ] && steps=10; [ $c = C4 ] && steps=8; [ $c = C1 -o $c = C2 ] && steps=60
      GSPLAT_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --config $c --steps $steps --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$(basename $(dirname $lib))', '$c', 'ms/step %.4f' % d['ms_per_step'], 'fwd %.4f bwd %.4f lists %.4f' % (s['composite_fwd'], s['composite_bwd'], s['tile_sort']))"
    done
  done
done
