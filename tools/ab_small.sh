#!/bin/bash
# C1 / C2 frames with 1 wave per tile against the automatic choice (gs_config.tile_parts), same box, interleaved.
for rep in 1 2 3; do
  for parts in 1 0; do
    for c in C1 C2; do
      GSPLAT_TILE_PARTS=$parts timeout -k 10 200 python3 bench.py --config $c --steps 60 --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('tile_parts=$parts', '$c', 'ms/frame %.4f' % d['ms_per_step'], 'fwd %.4f bwd %.4f' % (s['composite_fwd'], s['composite_bwd']))"
    done
  done
done
