#!/bin/bash
# round 5, third GPU call: why is the static schedule slower?  tile clocks of both schedules, 4 vs 5 waves per SIMD builds, forced rounds
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r05c
timeout -k 10 600 python -m pytest tests/test_gpu_bin3.py tests/test_gpu_parity.py tests/test_gpu_caps.py tests/test_gpu_rounds.py -q -m gpu -x > gpurun_out/${tag}_tests.log 2>&1; echo rc=$? >> gpurun_out/${tag}_tests.log; tail -3 gpurun_out/${tag}_tests.log
grep -q "rc=0" gpurun_out/${tag}_tests.log || { grep -n "Error\|assert\|FAILED" gpurun_out/${tag}_tests.log | head -30; exit 1; }
timeout -k 10 300 python3 tools/tile_tail.py C3 gpurun_out/${tag}_tile_tail_static_C3.json > gpurun_out/${tag}_tt_static.log 2>&1; echo "tt static rc=$?"
GSPLAT_SCHED_ROUNDS=1 timeout -k 10 300 python3 tools/tile_tail.py C3 gpurun_out/${tag}_tile_tail_dynamic_C3.json > gpurun_out/${tag}_tt_dynamic.log 2>&1; echo "tt dynamic rc=$?"
run() { # lib sched_rounds config
  GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/$1/libgsplat_hip.so timeout -k 10 300 python3 bench.py --config $3 --steps 20 --warmup 5 --sched-rounds $2 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$1 sched_rounds=$2', '$3', 'rounds used', d['config'].get('sched_rounds'), 'ms/frame %.4f' % d['ms_per_step'], 'fwd %.4f bwd %.4f lists %.4f' % (s['composite_fwd'], s['composite_bwd'], s['tile_sort']), 'dom %.4f' % d['roofline']['avg_ms'])"
}
for rep in 1 2; do
  run lib 1 C3; run lib 0 C3; run lib 3 C3; run lib 5 C3; run lib_r5 0 C3; run lib_r5 3 C3
done 2>&1 | tee gpurun_out/${tag}_ab.log
python3 - <<'PY'
import json
for k in ("static","dynamic"):
    d=json.load(open(f"gpurun_out/r05c_tile_tail_{k}_C3.json"))
    print(k, "rounds", d.get("sched_rounds"), "clock", d.get("clock_mhz"))
    for v in ("fwd_v30","bwd_v30"):
        a=d[v]; print("  ",v, {x:a[x] for x in ("span_us","mean_ms_of_8_launches","peak_waves_in_flight","mean_over_peak","simd_time_share_by_resident_waves_0_to_8","simd_entries_per_us_by_resident_waves_0_to_8","tile_duration_us_percentiles","evaluated_per_simd_max_over_mean","simd_finish_spread_us")})
PY
