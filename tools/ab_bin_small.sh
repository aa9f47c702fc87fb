#!/bin/bash
# C1 (and C2 as the control: beyond the small path's limits) with gs_bin's small-frame path (bin_path 0) against the two-level path (3), same box, interleaved.
for rep in 1 2 3; do
  for bp in 3 0; do
    for c in C1 C2; do
      timeout -k 10 200 python3 bench.py --config $c --bin-path $bp --steps 100 --warmup 10 --no-cpu-baseline --no-literal --no-clustered --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('bin_path=$bp', '$c', 'path', d['config']['bin_path_of_frame'], 'ms/frame %.4f' % d['ms_per_step'], ' '.join('%s %.4f' % (k, s[k]) for k in ('preprocess','depth_sort','count_scan','tile_sort','composite_fwd','composite_bwd') if k in s))"
    done
  done
done
