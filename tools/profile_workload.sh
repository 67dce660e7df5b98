#!/bin/bash
# rocprofv3 kernel-trace stats for a secondary bench workload (zoo | mines | doors); run on the GPU box via gpurun.
set -e
WL=${1:-zoo}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --async-streams 0 --open-loop-chunk 0 --workload $WL > $OUT/trace.log 2>&1
grep -h '"metric"' $OUT/trace.log | tail -n 1 > $OUT/bench_line.json || true
find $OUT -name "*kernel_stats.csv" | head -n 2
