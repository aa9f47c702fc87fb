#!/bin/bash
# Build the library of a git revision beside the working tree's (same-box A/B of whole libraries, tools/ab_libs_c3.sh):
#   tools/build_ref_lib.sh [REV=HEAD]  ->  gaussiansplat_amd/lib_ref/libgsplat_hip.so
set -e
REV=${1:-HEAD}
cd "$(dirname "$0")/.."
D=tools/abl/ref_src
rm -rf $D; mkdir -p $D/x/y/csrc $D/x/include gaussiansplat_amd/lib_ref
for f in $(git ls-tree --name-only $REV gaussiansplat_amd/csrc/); do git show $REV:$f > $D/x/y/csrc/$(basename $f); done
git show $REV:include/gsplat.h > $D/x/include/gsplat.h
cd $D/x/y/csrc
for f in gs_preprocess gs_preprocess2d; do /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off -c $f.hip -o $f.o & done
for f in $(ls *.hip | sed 's/\.hip$//' | grep -v '^gs_preprocess$\|^gs_preprocess2d$'); do /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -c $f.hip -o $f.o & done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../../../../../gaussiansplat_amd/lib_ref/libgsplat_hip.so *.o -ldl
echo built gaussiansplat_amd/lib_ref/libgsplat_hip.so from $REV
