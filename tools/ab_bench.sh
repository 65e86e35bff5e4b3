#!/bin/bash
# A/B of whole bench steps on the GPU box: default library vs libcetkmc_hip_alt.so, alternating.  Usage: bash tools/ab_bench.sh [rounds]
R=${1:-2}
ALT=$GRAFT_REPO_ROOT/cet-driven-simulation-for-3d-printing-am-kmc-approach_amd/csrc/libcetkmc_hip_alt.so
ARGS="--steps 600 --warmup 50 --no-cpu-baseline --no-incremental --no-mode-b --no-phases"
for r in $(seq $R); do
  echo -n "base ms/step: "; python3 $GRAFT_REPO_ROOT/bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  echo -n "alt  ms/step: "; CETKMC_LIB=$ALT python3 $GRAFT_REPO_ROOT/bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
