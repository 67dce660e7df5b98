#!/bin/bash
# rocprofv3 kernel stats of the global_view kernels for a library variant: tools/gv_prof.sh TAG [lib.so] [workload]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; LIB=$2; WL=${3:-doors}
[ -n "$LIB" ] && export NPP_AMD_LIB=$ROOT/$LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_gv_$TAG -- python3 $ROOT/tools/gv_time.py $WL $4 2>&1 | grep global_view
f=$(ls $ROOT/gpurun_out/prof_gv_$TAG/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gv_cells' in r['Name'] or 'global_view' in r['Name']: print("   ", r['Name'][28:52], r['Calls'], "avg us", round(float(r['AverageNs'])/1e3,1))
PY
