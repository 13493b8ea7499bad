#!/bin/bash
# usage (on the GPU box): tools/pmc_secondary.sh <outdir> <config> "<counters...>"  -- one rocprofv3 --pmc pass of tools/secondary_bench.py
out=$1; cfg=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $@ --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/$out -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/secondary_bench.py $cfg > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob("gpurun_out/$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if "dlm::" in k:
        print(k, {c: round(v / max(1, cnt[(k, c)]), 1) for c, v in acc[k].items()}, "launches", max(cnt[(k, c)] for c in acc[k]))
PY
