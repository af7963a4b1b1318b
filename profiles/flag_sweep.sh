#!/bin/bash
# kernel_ms.sh over every library in rgb-d-slam_amd/lib/exp (build them with profiles/build_tu_variant.sh), default library first and last
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== default"; bash $R/profiles/kernel_ms.sh "$@"
for so in $R/rgb-d-slam_amd/lib/exp/libcape_*.so; do
  echo "== $(basename $so .so | sed s/libcape_//)"
  CAPE_HIP_LIB=$so bash $R/profiles/kernel_ms.sh "$@"
done
echo "== default"; bash $R/profiles/kernel_ms.sh "$@"
