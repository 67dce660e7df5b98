#!/bin/bash
# Pinned-variant sweep of one library on the bench workloads (GPU box):  tools/variant_sweep.sh LIB [workloads...]
LIB=$1; shift
WL=${@:-c0 mines doors c3mixed}
mkdir -p gpurun_out
for w in $WL; do
  for v in 0 1 2; do
    NPP_AMD_LIB=$(pwd)/$LIB timeout -k 10 150 python bench.py --workload $w --step-variant $v --steps 600 --warmup 50 --no-cpu-baseline --async-streams 0 \
      --open-loop-chunk 0 > gpurun_out/vs.json 2> gpurun_out/vs.err || { echo "FAILED $w $v"; tail -3 gpurun_out/vs.err; exit 1; }
    python - "$w" "$v" gpurun_out/vs.json <<'PY'
import json, sys
l = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
u = l["launch_us"]
print("%-8s variant %s  %7.2f M  mean %6.1f p50 %6.1f p95 %6.1f" % (sys.argv[1], sys.argv[2], l["value"] / 1e6, u["mean"], u["p50"], u["p95"]))
PY
  done
done
