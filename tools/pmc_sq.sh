#!/bin/bash
# SQ-side counters of a tool run, per kernel (one pass, 8 SQ slots).  Usage on the GPU box: bash tools/pmc_sq.sh <python tool> [args]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export REPS=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/$1 ${@:2} > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$R/gpurun_out/pmc_sq/*/*counter_collection.csv"), key=lambda p: __import__("os").path.getmtime(p))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-40:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"]):
    n = calls[k]
    if n < 10: continue
    a = acc[k]; w = a["SQ_WAVES"] / n
    print(f"{k:42s} waves {w:8.0f}  per wave: valu {a['SQ_INSTS_VALU']/a['SQ_WAVES']:7.0f} salu {a['SQ_INSTS_SALU']/a['SQ_WAVES']:6.0f} "
          f"vmem_rd {a['SQ_INSTS_VMEM_RD']/a['SQ_WAVES']:5.0f} vmem_wr {a['SQ_INSTS_VMEM_WR']/a['SQ_WAVES']:5.0f}  "
          f"wave_cycles(quad) {a['SQ_WAVE_CYCLES']/a['SQ_WAVES']:8.0f} wait_any {a['SQ_WAIT_ANY']/a['SQ_WAVES']:8.0f} active {a['SQ_ACTIVE_INST_ANY']/a['SQ_WAVES']:7.0f}")
PY
