#!/bin/bash
# Same-box A/B of library variants (payload row size x work classes of the launch order): composite kernel time and HBM traffic.
#   python -m gaussiansplat_amd.build --tag p48b16 -DGS_LPT_BUCKETS=16 ; ... --tag p64b32 -DGS_PAYLOAD_QUADS=4 ; ... ; tools/ab_traffic.sh
export TMPDIR=/tmp
B="python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-literal --no-train-iteration --no-c4-anchor"
for lib in lib lib_p48b16 lib_p64b32 lib_p64b16 lib; do
  export GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/$lib/libgsplat_hip.so
  O=$PWD/gpurun_out/abt_$lib; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- $B > $O/f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- $B > $O/w.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --output-format csv -d $O/t -- $B > $O/t.log 2>&1
  python3 tools/pmc_summary.py $O/s.json $O/f $O/w $O/t --match composite > /dev/null
  python3 - "$lib" "$O/s.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
for k,v in d["kernels"].items():
    if "<true" in k:
        print(sys.argv[1], k[5:22], "fetch_x2 %.0f MB  write %.0f MB  hit %.2f  t %.0f us" % (v["hbm_bytes_fetch_x2"]/1e6, v["hbm_bytes_write"]/1e6, v["TCC_HIT_sum"]/(v["TCC_HIT_sum"]+v["TCC_MISS_sum"]), v["mean_ns_under_pmc"]/1e3))
PY
  $B 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$lib', 'ms_per_step', round(d['ms_per_step'],4), 'bwd', round(d['roofline']['avg_ms'],4))"
  rm -rf $O
done
