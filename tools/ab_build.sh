#!/bin/bash
# Build an alternative libcetkmc_hip_${ALT_TAG:-alt}.so with extra compiler flags (e.g. -DCETKMC_SWEEP_UNROLL=4) for
# tools/ab_run.sh.  Usage (in the build container): bash tools/ab_build.sh -DCETKMC_SWEEP_UNROLL=4
cd "$(dirname "$0")/../cet-driven-simulation-for-3d-printing-am-kmc-approach_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -I/opt/rocm/include \
    "$@" cetkmc_hip.hip -o libcetkmc_hip_${ALT_TAG:-alt}.so -ldl
