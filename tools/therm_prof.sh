#!/bin/bash
# Kernel-trace A/B of the temperature kernels inside the real loop (config 3, laser + latent heat):
# rocprofv3 --kernel-trace --stats of bench.py --steps 200 for each "variant[:planes]" given; prints the k_thermal* averages.
# Usage on the GPU box: bash tools/therm_prof.sh 1 3:8 3:16   (parse the kernel_stats.csv files under gpurun_out/thprof_*)
OUT=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%:*}; ni=${spec#*:}
  opts="--set-option thermal_variant=$v"
  if [ "$ni" != "$spec" ]; then
    if [ "$v" = "1" ] || [ "$v" = "3" ] || [ "$v" = "5" ]; then opts="$opts --set-option thermal_planes_per_block16=$ni"; else opts="$opts --set-option thermal_planes_per_block=$ni"; fi
  fi
  d=$OUT/thprof_${v}_${ni}
  rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $B --L ${THERM_L:-256} --steps 200 --warmup 20 --no-cpu-baseline --no-incremental --no-mode-b --no-phases --no-recompute --no-512 --no-live-traffic $opts > /dev/null 2>&1
  f=$(ls $d/*/*kernel_stats.csv | head -1)
  echo "== variant $spec"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_thermal", "k_rate_table", "k_sweep_stream")):
        print("   %-60s calls %s avg_us %.2f min_us %.2f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
