#!/bin/bash
# counters for one conv shape through the microbenchmark. usage: bash tools/pmc_conv.sh <tag> <shape> "<counters>" [variant]
set -e
TAG=$1; SHAPE=$2; CTRS=$3; VAR=${4:-plain}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmcc_${TAG} -- python3 tools/bench_conv.py --shapes $SHAPE --variants $VAR --iters 5 > gpurun_out/pmcc_${TAG}.log 2>&1
f=$(find gpurun_out/pmcc_${TAG} -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
tot=collections.defaultdict(float);n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv3x3' in r['Kernel_Name']:
        tot[r['Counter_Name']]+=float(r['Counter_Value']);n[r['Counter_Name']].add(r['Dispatch_Id'])
for k in tot: print(k, tot[k]/len(n[k]))
PY
