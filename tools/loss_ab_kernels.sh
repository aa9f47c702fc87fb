#!/bin/bash
# Per-kernel average of the loss kernels for several library builds (rocprofv3 --stats of tools/loss_prof.py).
export TMPDIR=/tmp
for d in "$@"; do
  O=$PWD/gpurun_out/loss_abk_$d; rm -rf $O
  (cd /tmp && GSPLAT_HIP_LIB=$OLDPWD/gaussiansplat_amd/$d/libgsplat_hip.so rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $OLDPWD/tools/loss_prof.py > /dev/null 2>&1)
  echo "== $d"; find $O -name "*kernel_stats.csv" | xargs grep ssim | awk -F, '{printf "   %-45s %8.1f us\n", $1, $4/1000}'
done
