#!/bin/bash
# On the GPU box: tools/pmc_wr.sh <name> <kernel-substring> <bench args...> -- write-request counters of one kernel
name=$1; kern=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$name; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum WRITE_SIZE --kernel-trace -d $O/p1 -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary "$@" > $O/p1.log 2>&1
cd $R
python3 - "$O" "$kern" <<'PY'
import csv, glob, sys, collections
O, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); cnt = collections.Counter()
for fn in glob.glob(O + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in acc: print(f"{k:28s} {acc[k] / cnt[k]:16.0f}  (launches {cnt[k]})")
if not acc: print("no rows; tail of log:"); print(open(O + "/p1.log").read()[-1500:])
PY
