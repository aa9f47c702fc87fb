#!/bin/bash
# Same-box A/B of whole libraries on the small configs:  tools/ab_libs_small.sh libA libB ...
set -e -o pipefail
for rep in 1 2 3; do
  for lib in "$@"; do
    for c in C1 C2; do
      GSPLAT_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --config $c --no-cpu-baseline --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$(basename $(dirname $lib))', '$c', 'ms/frame %.4f' % d['ms_per_step'], 'fwd %.3f bwd %.3f' % (s['composite_fwd'], s['composite_bwd']))"
    done
  done
done
