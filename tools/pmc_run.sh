#!/bin/bash
# usage (on the GPU box): tools/pmc_run.sh <outdir> "<counters...>"   -- one rocprofv3 --pmc pass of the bench
# prints per-kernel counter sums (divide by waves x steps for per wave-step figures)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $@ --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/$out -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/$out/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if "sp16" in k or "tiled" in k or "svd" in k:
        print(k, {c: round(v / max(1, cnt[(k, c)]), 1) for c, v in acc[k].items()})
PY
