#!/bin/bash
# pinned step-kernel build 0 of two libraries on the three level sets
cd ${GRAFT_REPO_ROOT:-.}
for wl in c0 mines doors; do
  for lib in "$@"; do
    NPP_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $wl --steps 600 --warmup 50 --step-variant 0 --no-cpu-baseline --async-streams 0 --open-loop-chunk 0 > gpurun_out/v0ab.log 2>&1 || { tail -3 gpurun_out/v0ab.log; exit 1; }
    python - "$wl" "$lib" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/v0ab.log") if l.startswith("{")][-1])
print("%-6s %-28s %7.2f M  mean %.1f p50 %.1f p95 %.1f" % (sys.argv[1], sys.argv[2], d["value"] / 1e6, d["launch_us"]["mean"], d["launch_us"]["p50"], d["launch_us"]["p95"]))
PY
  done
done
