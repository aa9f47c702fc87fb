#!/bin/bash
# Builds the standalone gfx950 microbenchmarks next to their sources (run them on the GPU box: gpurun -- tools/valu_ubench3).
set -e
cd "$(dirname "$0")"
for t in lds_atomic_order valu_ubench valu_ubench2 valu_ubench3 valu_ubench4 graph_probe wg_dispatch_probe; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $t.hip -o $t
done
