#!/bin/bash
# Build tools/variants/libabl_<NAME>.so = the product sources + extra compiler flags
# (ablation / diagnostic builds; load one with BB_LIB=$PWD/tools/variants/libabl_<NAME>.so).
#   usage: tools/build_variant.sh NAME [flags...]      e.g.  tools/build_variant.sh TRACE -DBB_WAVE_TRACE
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p tools/variants
name=$1; shift
SRC="blueberry_amd/csrc/bb_api.cpp blueberry_amd/csrc/bb_comm.cpp blueberry_amd/csrc/bb_solver.hip blueberry_amd/csrc/bb_band.hip blueberry_amd/csrc/bb_contactmap.hip blueberry_amd/csrc/bb_misc.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
    -Wno-unused-value -Wno-unused-result -fno-slp-vectorize -Iinclude -Iblueberry_amd/csrc "$@" \
    -o tools/variants/libabl_$name.so $SRC -ldl
echo "$*" > tools/variants/libabl_$name.flags
echo "built tools/variants/libabl_$name.so ($*)"
