#!/bin/bash
# A/B of several builds of the library on the bench workloads (run on the GPU box):
#   WL="c0 doors" STEPS=1000 tools/ab_round3.sh LIB_A LIB_B ...
# Each line: workload, library, env-steps/s, launch mean / p50 / p95, variant.
WL=${WL:-c0 doors mines}
STEPS=${STEPS:-600}
mkdir -p gpurun_out
for w in $WL; do
  for lib in "$@"; do
    NPP_AMD_LIB=$(pwd)/$lib timeout -k 10 150 python bench.py --workload $w --steps $STEPS --warmup 50 --no-cpu-baseline --async-streams 0 --open-loop-chunk 0 \
      > gpurun_out/ab_$w.json 2> gpurun_out/ab_$w.err || { echo "FAILED $w $lib"; tail -3 gpurun_out/ab_$w.err; exit 1; }
    python - "$w" "$lib" gpurun_out/ab_$w.json <<'PY'
import json, sys
l = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
u = l["launch_us"]
print("%-8s %-32s %7.2f M  mean %6.1f p50 %6.1f p95 %6.1f max %6.1f variant %d" % (sys.argv[1], sys.argv[2], l["value"] / 1e6, u["mean"], u["p50"], u["p95"], u["max"], l["step_variant"]["variant"]))
PY
  done
done
