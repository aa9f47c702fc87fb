#!/bin/bash
# round 5, second GPU call: the suite with the static schedule on, then same-box A/B of sched_rounds 1 (rounds 2-4) vs automatic
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r05b}
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/${tag}_tests.log 2>&1; echo rc=$? >> gpurun_out/${tag}_tests.log; tail -5 gpurun_out/${tag}_tests.log
grep -q "rc=0" gpurun_out/${tag}_tests.log || { grep -n "Error\|assert\|FAILED" gpurun_out/${tag}_tests.log | head -30; exit 1; }
for rep in 1 2 3; do
  for sr in 1 0; do
    for c in C3 C5; do
      steps=20; [ $c = C5 ] && steps=10
      timeout -k 10 300 python3 bench.py --config $c --steps $steps --warmup 5 --sched-rounds $sr --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('sched_rounds=$sr', '$c', 'rounds used', d['config'].get('sched_rounds'), 'ms/frame %.4f' % d['ms_per_step'], 'fwd %.4f bwd %.4f lists %.4f' % (s['composite_fwd'], s['composite_bwd'], s['tile_sort']), 'dom %.4f' % d['roofline']['avg_ms'])"
    done
  done
done 2>&1 | tee gpurun_out/${tag}_ab_sched_rounds.log
