#!/bin/bash
set -e -o pipefail
export GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/lib_exp/libgsplat_hip.so
for rep in 1 2; do
for g in 256 512 1024; do
  for c in C5 C3; do
  GS_L1_G=$g timeout -k 10 300 python3 bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --no-train-iteration --no-c4-anchor --no-literal 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('G=$g $c ms/frame %.4f' % d['ms_per_step'], 'count_scan %.3f tile_sort %.3f' % (s['count_scan'], s['tile_sort']))"
  done
done
done
