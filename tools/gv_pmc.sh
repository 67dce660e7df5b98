#!/bin/bash
# SQ counters of the global_view kernels: tools/gv_pmc.sh TAG lib.so workload n "COUNTER ..."
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; LIB=$2; WL=${3:-doors}; N=${4:-8192}; CTR=${5:-"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"}
[ -n "$LIB" ] && [ "$LIB" != "-" ] && export NPP_AMD_LIB=$ROOT/$LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $ROOT/gpurun_out/pmc_gv_$TAG -- python3 $ROOT/tools/gv_time.py $WL $N > /dev/null 2>&1
f=$(ls $ROOT/gpurun_out/pmc_gv_$TAG/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k='cells' if 'gv_cells' in r['Kernel_Name'] else ('gview' if 'global_view' in r['Kernel_Name'] else None)
    if not k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k in agg:
    print(k, {c: round(v/len(cnt[k])) for c,v in agg[k].items()})
PY
