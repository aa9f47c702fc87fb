#!/bin/bash
# Same-box A/B of tagged library builds at C5 (5 M gaussians, 4K): frame time and the binning stages.
#   tools/ab_c5.sh lib_ref lib
set -e -o pipefail
for rep in 1 2; do
  for d in "$@"; do
    GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/$d/libgsplat_hip.so timeout -k 10 300 python3 bench.py --config C5 --steps 6 --warmup 2 --no-cpu-baseline --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$d', 'C5 ms/frame %.4f' % d['ms_per_step'], ' '.join('%s %.3f' % (k, v) for k, v in s.items()))"
  done
done
