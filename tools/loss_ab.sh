#!/bin/bash
# Same-box timing of the loss kernels for several library builds (directory names under gaussiansplat_amd/), interleaved.
for rep in 1 2 3; do
  for d in "$@"; do
    echo -n "$d "; GSPLAT_HIP_LIB=$PWD/gaussiansplat_amd/$d/libgsplat_hip.so python3 tools/loss_prof.py 2>/dev/null | python3 -c "import json,sys; print('%.1f us' % json.loads(sys.stdin.read())['us_per_call'])"
  done
done
