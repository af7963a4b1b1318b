#!/bin/bash
# A/B builds of libcape_hip.so for kernel experiments: rgb-d-slam_amd/lib/exp/libcape_<name>.so, loaded through
# CAPE_HIP_LIB (profiles/sweep.py "CAPE_HIP_LIB=... [bench args]").   usage: build_variant.sh <name> [-DFOO=1 ...]
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME=$1; shift
mkdir -p "$ROOT/rgb-d-slam_amd/lib/exp"
cd "$ROOT/rgb-d-slam_amd/csrc"
TAG=$NAME
# A1 is a translation unit of its own, compiled without the SLP vectoriser like the Makefile does
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function \
    -fno-slp-vectorize "$@" -c -o ../lib/exp/cape_cell_moments.$TAG.o cape_cell_moments.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function \
    "$@" -shared -o ../lib/exp/libcape_$NAME.so \
    ../lib/exp/cape_cell_moments.$TAG.o cape_api.hip cape_cell_fit.hip cape_grow.hip cape_resume.hip cape_polygon.hip cape_debug.hip cape_rectify.hip cape_match.hip cape_match_polygon.hip cape_gather.hip -ldl
echo built lib/exp/libcape_$NAME.so "$@"
