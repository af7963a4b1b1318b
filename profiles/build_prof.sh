#!/bin/bash
# Builds libcape_hip.so + the host mirror, and a -DCAPE_B_PROFILE twin (rgb-d-slam_amd/lib/exp/libcape_prof.so) that
# profiles/grow_phases.py loads through CAPE_HIP_LIB.  Extra -D flags for the twin: ./build_prof.sh -DFOO=1
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
make -C "$ROOT/rgb-d-slam_amd/csrc" all host 2>&1 | grep -E "warning|error" || true
mkdir -p "$ROOT/rgb-d-slam_amd/lib/exp"
cd "$ROOT/rgb-d-slam_amd/csrc"
TAG=prof
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function \
    -fno-slp-vectorize -DCAPE_B_PROFILE "$@" -c -o ../lib/exp/cape_cell_moments.$TAG.o cape_cell_moments.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function \
    -DCAPE_B_PROFILE "$@" -shared -o ../lib/exp/libcape_prof.so \
    ../lib/exp/cape_cell_moments.$TAG.o cape_api.hip cape_cell_fit.hip cape_grow.hip cape_resume.hip cape_polygon.hip cape_debug.hip cape_rectify.hip cape_match.hip cape_match_polygon.hip cape_gather.hip -ldl
ls -la "$ROOT/rgb-d-slam_amd/lib/libcape_hip.so" "$ROOT/rgb-d-slam_amd/lib/exp/libcape_prof.so" | awk '{print $6,$7,$8,$9}'
