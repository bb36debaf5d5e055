#!/bin/bash
# Whole-build A/B variants: tools/build_ab.sh NAME -DFOO=1 ...  ->  open3d_slam_private_amd/lib_ab/libNAME.so  (load with O3D_REG_LIB)
set -e
cd "$(dirname "$0")/../open3d_slam_private_amd/csrc"
name=$1; shift
mkdir -p ../lib_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-result \
  -DO3D_MATCH_WAVES=5 -DO3D_SEARCH_WAVES=4 -Wno-macro-redefined "$@" -shared -o ../lib_ab/lib$name.so reg_core.hip
