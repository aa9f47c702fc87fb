#!/bin/bash
# Same-box A/B of whole libraries at C3 (frame + survey stages), interleaved three times:  tools/ab_libs_c3.sh libA libB ...
set -e -o pipefail
for rep in 1 2 3; do
  for lib in "$@"; do
    GSPLAT_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --config C3 --no-cpu-baseline --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$(basename $(dirname $lib))', 'ms/frame %.4f' % d['ms_per_step'], ' '.join('%s %.3f' % (k[:10], v) for k, v in s.items() if v), 'dom %.4f' % d['roofline']['avg_ms'])"
  done
done
