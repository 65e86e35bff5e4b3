#!/bin/bash
# Round profile: plain bench line, rocprofv3 kernel stats, and HBM-traffic PMC passes (separate runs).
# Usage on the GPU box: bash tools/profile_round.sh r01 ; results under gpurun_out/<tag>_*
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-incremental --no-mode-b --no-phases --no-recompute > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-incremental --no-mode-b --no-phases --no-recompute > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-incremental --no-mode-b --no-phases --no-recompute > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_modes -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 40 --no-cpu-baseline > $OUT/${TAG}_bench_modes_under_rocprof.json 2> /dev/null
echo done
