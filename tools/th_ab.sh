#!/bin/bash
# A/B of the temperature kernel: in-tree library against csrc/libcetkmc_hip_alt.so (tools/ab_build.sh), per-step thermal share of bench.py's phase table
R=$GRAFT_REPO_ROOT
for lib in "" "$R/cet-driven-simulation-for-3d-printing-am-kmc-approach_amd/csrc/libcetkmc_hip_alt.so"; do
  for rep in 1 2; do
    if [ -z "$lib" ]; then unset CETKMC_LIB; else export CETKMC_LIB=$lib; fi
    python $R/bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-recompute --no-mode-b --no-incremental > $R/gpurun_out/th_ab.json 2>/dev/null
    python -c "
import json; d=json.load(open('$R/gpurun_out/th_ab.json')); print('lib', '${lib:-in-tree}'[-12:], d['value'], d['phases']['thermal_us_per_step'])"
  done
done
