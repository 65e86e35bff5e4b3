#!/bin/bash
# kernel-trace of the Mode B leg for the default build and the alternative builds given (tags)
OUT=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" = "default" ]; then unset CETKMC_LIB; else export CETKMC_LIB=$GRAFT_REPO_ROOT/cet-driven-simulation-for-3d-printing-am-kmc-approach_amd/csrc/libcetkmc_hip_$tag.so; fi
  d=$OUT/touchprof_$tag
  rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $B --steps 100 --warmup 20 --no-cpu-baseline --no-incremental --no-phases --no-recompute --no-512 --no-live-traffic > /dev/null 2>&1
  f=$(ls $d/*/*kernel_stats.csv | head -1)
  echo "== build $tag"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_domain", "k_super", "k_interface_part", "k_ifc_relist")):
        print("   %-50s calls %s avg_us %.2f min_us %.2f max_us %.2f" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
