#!/bin/bash
# Sweep of the halo-bin edge (O3D_HALO_RATIO, units of the brick-table bin edge) on a C2- or C3-shaped registration.
# Usage: bash tools/sweep_halo.sh [n_src n_tgt]
NS=${1:-100000}; NT=${2:-1000000}
for r in 1.25 1.5 1.75 2.0; do echo "ratio $r"; O3D_HALO_RATIO=$r python tools/tools_reg1.py $NS $NT 5 | tail -1; done
