#!/bin/bash
# Cost-split launch experiment (NPP_STEP_SPLIT=percent:variant_heavy:variant_light; GPU box): bench lines with and without.
mkdir -p gpurun_out
run() {
  NPP_STEP_SPLIT=$2 timeout -k 10 150 python bench.py --workload $1 --steps 1000 --warmup 50 --no-cpu-baseline --async-streams 0 --open-loop-chunk 0 \
    > gpurun_out/sp.json 2> gpurun_out/sp.err || { echo "FAILED $1 $2"; tail -3 gpurun_out/sp.err; exit 1; }
  python - "$1" "${2:-none}" gpurun_out/sp.json <<'PY'
import json, sys
l = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
u = l["launch_us"]
print("%-8s split %-8s %7.2f M  ms/step %.4f  launch mean %6.1f p50 %6.1f p95 %6.1f  variant %d" % (sys.argv[1], sys.argv[2], l["value"] / 1e6, l["ms_per_step"], u["mean"], u["p50"], u["p95"], l["step_variant"]["variant"]))
PY
}
for w in c0 c3mixed doors; do
  run $w ""
  for s in 5:0:1 12:0:1 25:0:1 12:2:1; do run $w $s; done
done
