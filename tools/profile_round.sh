#!/bin/bash
# Round profile: plain bench line, rocprofv3 kernel stats, and HBM-traffic PMC passes (separate runs: gpurun refuses --pmc
# combined with trace domains other than the kernel trace).
# Usage on the GPU box: bash tools/profile_round.sh r03 [parts]; results under gpurun_out/<tag>_*
#   parts (default "bench stats pmc modes 512"): bench = plain default bench line; stats = kernel trace of the full-sweep loop;
#   pmc = FETCH_SIZE / WRITE_SIZE passes of the same loop; modes = kernel trace incl. incremental + Mode B;
#   512 = kernel trace + PMC passes of the 512^3 single-GPU loop (<tag>_512_*: working set 1.2 GB, no Infinity-Cache residency)
TAG=${1:-r03}
PARTS=${2:-"bench stats pmc modes 512"}
OUT=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
LOOP="--no-cpu-baseline --no-incremental --no-mode-b --no-phases --no-recompute --no-512 --no-live-traffic"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has bench; then python3 $B > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench done"; fi
if has stats; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $B --steps 400 --warmup 40 $LOOP > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null; echo "stats done"; fi
if has pmc; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $B --steps 40 --warmup 4 $LOOP > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $B --steps 40 --warmup 4 $LOOP > /dev/null 2>&1; echo "pmc done"; fi
if has modes; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_modes -- python3 $B --steps 400 --warmup 40 --no-cpu-baseline --no-512 --no-live-traffic > $OUT/${TAG}_bench_modes_under_rocprof.json 2> /dev/null; echo "modes done"; fi
if has modespmc; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_modes_pmc_fetch -- python3 $B --steps 40 --warmup 4 --no-cpu-baseline --no-512 --no-live-traffic --no-phases --no-recompute --no-incremental > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_modes_pmc_write -- python3 $B --steps 40 --warmup 4 --no-cpu-baseline --no-512 --no-live-traffic --no-phases --no-recompute --no-incremental > /dev/null 2>&1; echo "modespmc done"; fi
if has 512; then
  T5=${TAG}_512
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${T5}_stats -- python3 $B --L 512 --steps 60 --warmup 6 $LOOP > $OUT/${T5}_bench_under_rocprof.json 2> /dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${T5}_pmc_fetch -- python3 $B --L 512 --steps 20 --warmup 2 $LOOP > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${T5}_pmc_write -- python3 $B --L 512 --steps 20 --warmup 2 $LOOP > /dev/null 2>&1; echo "512 done"; fi
echo done
