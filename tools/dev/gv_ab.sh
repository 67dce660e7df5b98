#!/bin/bash
# global_view A/B of library builds: stand-alone launch time, then inside the full-obs step (serial and overlapped)
cd ${GRAFT_REPO_ROOT:-.}
for lib in "$@"; do
  NPP_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/gv_time.py doors 2>&1 | tail -1
  NPP_AMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload doors --full-obs --steps 300 --warmup 50 --step-variant 0 --no-cpu-baseline --obs-overlap 40 > gpurun_out/gvab.log 2>&1 || { tail -3 gpurun_out/gvab.log; exit 1; }
  python - $lib <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/gvab.log") if l.startswith("{")][-1])
print(sys.argv[1], "serial %.1f us" % (d["serial"]["ms_per_step"] * 1e3), "overlap", [("+".join(map(str,o["cuts_percent"])), round(o["ms_per_step"] * 1e3, 1)) for o in d["obs_overlap"]],
      {k: round(v["mean"], 1) for k, v in d["obs_kernels"].items() if k != "note"})
PY
done
