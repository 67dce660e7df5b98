#!/bin/bash
# Run on the GPU box (via gpurun): collects the rocprofv3 evidence for bench.py's roofline objects.
#   tools/profile_round.sh TAG [bench.py args: --workload c0 (default) | --workload mines --player-frame | --workload doors --full-obs]
#   1. --kernel-trace --stats        -> per-kernel average duration (must agree with bench.py's HIP-event figure)
#   2. --pmc FETCH_SIZE / WRITE_SIZE  -> HBM traffic, one counter per pass (TCC slots: MI355X_MICROARCH.md)
#   3. --pmc SQ_* counters            -> instruction mix / lane utilisation
# Output goes to gpurun_out/prof_<tag>/ ; tools/summarize_profiles.py condenses it into profiles/.
set -e
TAG=${1:-r02}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=${*:---workload c0}
BENCH="python3 $ROOT/bench.py --steps 300 --warmup 100 --no-cpu-baseline --async-streams 0 --open-loop-chunk 0 --obs-overlap= $ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1
grep -h '"metric"' $OUT/trace.log | tail -n 1 > $OUT/bench_line.json || true
find $OUT -name "*.csv" | head -n 40
