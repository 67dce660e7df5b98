#!/bin/bash
# doors --full-obs with npp_set_obs_overlap: one process per overlap spec (stream -> hardware queue mapping depends on creation order)
cd ${GRAFT_REPO_ROOT:-.}
for m in ${MASKS:-2}; do
 for spec in ${PCTS:-40}; do
  NPP_OBS_SIDE=$m timeout -k 10 300 python bench.py --workload doors --full-obs --steps 300 --warmup 50 --step-variant 0 --no-cpu-baseline --obs-overlap $spec > gpurun_out/ov_sweep_$m.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ov_sweep_$m.log") if l.startswith("{")][-1])
print("mask $m serial %.1f us" % (d["serial"]["ms_per_step"]*1e3), [("+".join(map(str,o["cuts_percent"])), round(o["ms_per_step"]*1e3,1)) for o in d["obs_overlap"]])
PY
 done
done
