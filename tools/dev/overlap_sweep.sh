#!/bin/bash
# doors --full-obs with npp_set_obs_overlap, for several NPP_OBS_SIDE masks (which observation kernels get their own streams)
cd ${GRAFT_REPO_ROOT:-.}
for m in ${MASKS:-0 1 2}; do
  NPP_OBS_SIDE=$m timeout -k 10 300 python bench.py --workload doors --full-obs --steps 300 --warmup 50 --step-variant 0 --no-cpu-baseline --obs-overlap ${PCTS:-12,25,40} > gpurun_out/ov_sweep_$m.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ov_sweep_$m.log") if l.startswith("{")][-1])
print("mask $m serial %.1f us" % (d["ms_per_step"]*1e3), [(o["percent"], round(o["ms_per_step"]*1e3,1)) for o in d["obs_overlap"]])
PY
done
