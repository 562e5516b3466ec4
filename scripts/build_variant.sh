#!/bin/bash
# A library variant for same-box A/B runs (scripts/ab_bench.sh): one kernel file recompiled with extra -D flags, the other objects
# from the regular build.   usage: scripts/build_variant.sh <name> <file.hip> [-DMACRO=value ...]   ->  build_ab/<name>.so
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../orb-slam3-rust_amd/csrc"
make -s
OBJ=/tmp/orbx_variant_${NAME}_$$.o
EXTRA=""
if [ "$SRC" = "orb_kernels.hip" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"; fi   # as the Makefile builds that file
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-result \
    -Wno-unused-value -Wno-pass-failed $EXTRA "$@" -c "$SRC" -o "$OBJ"
mkdir -p ../../build_ab
OBJS=""
for f in orbx_api match_kernels orb_kernels ba_kernels bow_kernels euroc_io keyframe; do
  if [ "$f.hip" = "$SRC" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/$NAME.so $OBJS -lz -lpthread -ldl
rm -f "$OBJ"
echo build_ab/$NAME.so
