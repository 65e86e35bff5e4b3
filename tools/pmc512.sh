#!/bin/bash
# FETCH_SIZE / WRITE_SIZE + kernel time of the 512^3 sweep for alternative builds (csrc/libcetkmc_hip_<tag>.so; "base" = in-tree).
# Usage on the GPU box: bash tools/pmc512.sh base ni16 notail
OUT=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
LOOP="--L ${PMC_L:-512} --steps 20 --warmup 2 --no-cpu-baseline --no-incremental --no-mode-b --no-phases --no-recompute --no-512 --no-live-traffic"
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" = "base" ]; then unset CETKMC_LIB; else export CETKMC_LIB=$GRAFT_REPO_ROOT/cet-driven-simulation-for-3d-printing-am-kmc-approach_amd/csrc/libcetkmc_hip_$tag.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc512_${tag}_$c
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc512_${tag}_$c -- python3 $B $LOOP > /dev/null 2>&1
  done
  rm -rf $OUT/pmc512_${tag}_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc512_${tag}_stats -- python3 $B $LOOP > /dev/null 2>&1
  python3 - "$tag" <<'PY'
import csv, glob, os, statistics, sys
tag = sys.argv[1]
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc512_{tag}_{c}/*/*counter_collection.csv")[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_sweep_stream" in r["Kernel_Name"]]
    res[c] = statistics.median(v)
f = glob.glob(f"{out}/pmc512_{tag}_stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "k_sweep_stream" in r["Name"]:
        res["avg_us"] = float(r["AverageNs"]) / 1e3
print(f"{tag:8s} sweep avg {res.get('avg_us', 0):8.2f} us  2xFETCH {2 * res['FETCH_SIZE'] * 1024 / 1e6:9.1f} MB  WRITE {res['WRITE_SIZE'] * 1024 / 1e6:7.1f} MB", flush=True)
PY
done
