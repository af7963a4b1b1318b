#!/bin/bash
# A/B builds that differ ONLY in one translation unit's flags (the other objects are the default build's: run `make` first):
# usage: build_tu_variant.sh <name> <file.hip> [flags...]  -> rgb-d-slam_amd/lib/exp/libcape_<name>.so (load through CAPE_HIP_LIB)
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME=$1; SRC=$2; shift; shift
TU=$(basename $SRC .hip)
mkdir -p "$ROOT/rgb-d-slam_amd/lib/exp"
cd "$ROOT/rgb-d-slam_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c -o ../lib/exp/$TU.$NAME.o $TU.hip
OBJS=$(ls ../lib/obj/*.o | grep -v cyl_exact | grep -v "/$TU.o")
/opt/rocm/bin/hipcc $F -shared -o ../lib/exp/libcape_$NAME.so ../lib/exp/$TU.$NAME.o $OBJS -ldl
echo built lib/exp/libcape_$NAME.so $TU.hip "$@"
