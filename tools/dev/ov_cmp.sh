#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for args in "--step-variant 0 --steps 300 --warmup 50" "--steps 300 --warmup 50" "--steps 200 --warmup 20" "--step-variant 0 --steps 200 --warmup 20"; do
  timeout -k 10 300 python bench.py --workload doors --full-obs --no-cpu-baseline --obs-overlap 40 $args > gpurun_out/ovc.log 2>&1 || exit 1
  python - "$args" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ovc.log") if l.startswith("{")][-1])
print(sys.argv[1], "| serial %.1f" % (d["serial"]["ms_per_step"] * 1e3), "overlap %.1f" % (d["obs_overlap"][0]["ms_per_step"] * 1e3), "variant", d["step_variant"])
PY
done
