#!/bin/bash
# Build libblueberry_hip.so (gfx950) and the CPU oracle.  Used by
# __graft_entry__.build(); safe to run by hand.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
# -fno-slp-vectorize: the SLP vectorizer fuses the pair math of all 8 rows of a
# unit into long v_pk_* trees and spills ~500 B/lane in stress_grad_kernel;
# without it the kernel needs 104 VGPRs and no scratch.
SRC="blueberry_amd/csrc/bb_api.cpp blueberry_amd/csrc/bb_comm.cpp blueberry_amd/csrc/bb_solver.hip blueberry_amd/csrc/bb_band.hip blueberry_amd/csrc/bb_contactmap.hip blueberry_amd/csrc/bb_misc.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
    -Wno-unused-value -Wno-unused-result -fno-slp-vectorize \
    -Iinclude -Iblueberry_amd/csrc ${BB_EXTRA_FLAGS:-} \
    -o blueberry_amd/libblueberry_hip.so $SRC -ldl
make -s -C oracle
echo "built blueberry_amd/libblueberry_hip.so and oracle/libbb_oracle.so"
