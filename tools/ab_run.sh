#!/bin/bash
# A/B on the GPU box: default library vs libcetkmc_hip_alt.so, alternating, same box.  Usage: bash tools/ab_run.sh [rounds]
R=${1:-3}
ALT=$GRAFT_REPO_ROOT/cet-driven-simulation-for-3d-printing-am-kmc-approach_amd/csrc/libcetkmc_hip_alt.so
for r in $(seq $R); do
  echo -n "base: "; python3 $GRAFT_REPO_ROOT/tools/bench_variants.py 256 1 3 | tail -1
  echo -n "alt : "; CETKMC_LIB=$ALT python3 $GRAFT_REPO_ROOT/tools/bench_variants.py 256 1 3 | tail -1
done
