#!/bin/bash
set -e -o pipefail
for rep in 1 2; do
  for d in "$@"; do
    GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/$d/libgsplat_hip.so timeout -k 10 200 python3 bench.py --config C2 --no-cpu-baseline --no-train-iteration --no-c4-anchor --no-literal 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$d', 'C2 ms/frame %.4f' % d['ms_per_step'], 'fwd %.3f bwd %.3f' % (s['composite_fwd'], s['composite_bwd']))"
  done
done
