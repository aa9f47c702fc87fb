#!/bin/bash
# VGPRs / spills / occupancy of every kernel of one translation unit (hipcc's kernel-resource-usage remarks).
# usage: tools/kernel_regs.sh gs_composite.hip [extra -D flags]
src=gaussiansplat_amd/csrc/$1
case "$1" in gs_composite.hip|gs_loss.hip) noslp=-fno-slp-vectorize;; *) noslp=;; esac   # as gaussiansplat_amd/build.py compiles them
shift
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function $noslp "$@" -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur: rows[cur][m.group(1)] = int(m.group(2))
import subprocess
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    print("%-110s vgpr %3d sgpr %3d scratch %3d spill %d occ %d lds %d" % (name[:110], v.get("VGPRs", -1), v.get("TotalSGPRs", -1), v.get("ScratchSize [bytes/lane]", -1), v.get("VGPRs Spill", -1), v.get("Occupancy [waves/SIMD]", -1), v.get("LDS Size [bytes/block]", -1)))
'
