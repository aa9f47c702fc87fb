#!/bin/bash
# Same-box comparison of the working tree with round 3's final tree (tools/abl/r03_tree: `git archive b3184a7` + its library built by
# tools/build_ref_lib.sh b3184a7), whole bench lines, interleaved.   tools/ab_vs_r03.sh "C3 C4 C5 C2 C1"
CFGS=${1:-"C3 C4"}
R=$PWD
for rep in 1 2 3; do
  for tree in tools/abl/r03_tree .; do
    for c in $CFGS; do
      steps=20; [ $c = C5 ] && steps=10; [ $c = C4 ] && steps=8; [ $c = C1 -o $c = C2 ] && steps=60
      (cd $R/$tree && timeout -k 10 300 python3 bench.py --config $c --steps $steps --warmup 5 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$tree', '$c', 'ms/step %.4f' % d['ms_per_step'], 'Msplats/s %.1f' % d['value'], 'fwd %.4f bwd %.4f lists %.4f' % (s['composite_fwd'], s['composite_bwd'], s['tile_sort']))")
    done
  done
done
