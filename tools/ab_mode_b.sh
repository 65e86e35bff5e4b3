#!/bin/bash
# A/B of Mode B on one box: the in-tree library against csrc/libcetkmc_hip_alt.so (tools/ab_build.sh), each under
# rocprofv3 kernel stats; prints the per-kernel averages side by side.  GPU box only.
R=$GRAFT_REPO_ROOT
ALT=$(ls $R/cet-driven-simulation-for-3d-printing-am-kmc-approach_amd/csrc/libcetkmc_hip_alt.so)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_new -- python3 $R/tools/bench_mode_b.py 256 8 40 > $R/gpurun_out/ab_new.log 2>&1 || exit 1
export CETKMC_LIB=$ALT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_old -- python3 $R/tools/bench_mode_b.py 256 8 40 > $R/gpurun_out/ab_old.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
def stats(d):
    f = sorted(glob.glob(f"$R/gpurun_out/{d}/*/*kernel_stats.csv"))[-1]
    return {r["Name"].split("(")[0][-40:]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f))}
a, b = stats("ab_new"), stats("ab_old")
for k in a:
    if a[k][0] >= 100:
        print(f"{k:42s} new {a[k][1]:8.2f} us   alt {b.get(k, (0, float('nan')))[1]:8.2f} us")
PY
grep executed $R/gpurun_out/ab_new.log | head -2; grep executed $R/gpurun_out/ab_old.log | head -2
