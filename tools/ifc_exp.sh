cd /tmp && export TMPDIR=/tmp
for cfg in "0 2" "100 2" "100 1" "400 2"; do set -- $cfg; EVOLVE=$1 THERMAL=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ev_$1_$2 -- python3 $GRAFT_REPO_ROOT/tools/bench_variants.py 256 1 1 > $GRAFT_REPO_ROOT/gpurun_out/ev_$1_$2.log 2>&1; done
