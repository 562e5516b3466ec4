#!/bin/bash
# library variant with the solve kernel's phase stamps (debug only): build_ab/bast.so
set -e
cd "$(dirname "$0")/../orb-slam3-rust_amd/csrc"
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DORBX_SOLVE_STAMPS -Wno-unused-function -Wno-unused-result -Wno-unused-value -c ba_kernels.hip -o /tmp/ba_st.o
mkdir -p ../../build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/bast.so orbx_api.o match_kernels.o orb_kernels.o /tmp/ba_st.o bow_kernels.o euroc_io.o keyframe.o -lz -lpthread -ldl
