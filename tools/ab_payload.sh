set -e
B="python3 bench.py --steps 80 --warmup 8 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor"
for r in 1 2 3; do
  for lib in lib lib_p48; do
    GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/$lib/libgsplat_hip.so timeout -k 10 100 $B 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print('$lib', round(d['ms_per_step'],4), d['stage_ms']['composite_fwd'], d['stage_ms']['composite_bwd'], d['stage_ms']['preprocess'], round(d['roofline']['avg_ms'],4))"
  done
done
GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/lib_p48/libgsplat_hip.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_2d.py tests/test_gpu_cull.py -q -x -p no:cacheprovider 2>&1 | tail -2
