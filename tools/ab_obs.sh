#!/bin/bash
# A/B of library builds on the workloads with observation kernels / zoo kernels (GPU box):  tools/ab_obs.sh LIB_A LIB_B ...
mkdir -p gpurun_out
for cfg in "zoo" "mines --player-frame" "doors --full-obs"; do
  for lib in "$@"; do
    NPP_AMD_LIB=$(pwd)/$lib timeout -k 10 200 python bench.py --workload $cfg --steps 300 --warmup 50 --no-cpu-baseline --async-streams 0 --open-loop-chunk 0 \
      > gpurun_out/abo.json 2> gpurun_out/abo.err || { echo "FAILED $cfg $lib"; tail -3 gpurun_out/abo.err; exit 1; }
    python - "$cfg" "$lib" gpurun_out/abo.json <<'PY'
import json, sys
l = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
extra = {k: round(v["mean"], 1) for k, v in l.get("obs_kernels", {}).items() if k != "note"}
if "roofline_render" in l and not extra:
    extra = {"player_frame": round(l["roofline_render"]["avg_launch_us"], 1)}
print("%-22s %-30s %7.2f M  step mean %6.1f  %s" % (sys.argv[1], sys.argv[2], l["value"] / 1e6, l["launch_us"]["mean"], extra))
PY
  done
done
