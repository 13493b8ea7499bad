#!/bin/bash
# On the GPU box: tools/pmc_sq.sh <name> <kernel-substring> <bench args...> -- SQ instruction / cycle counters of one kernel (two passes)
name=$1; kern=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$name; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace -d $O/p1 -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary "$@" > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/p2 -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary "$@" > $O/p2.log 2>&1
cd $R
python3 - "$O" "$kern" <<'PY'
import csv, glob, sys, collections
O, kern = sys.argv[1], sys.argv[2]
for p in ("p1", "p2"):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for fn in glob.glob(O + f"/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    for k in acc: print(f"{k:24s} {acc[k] / cnt[k]:16.0f}  (launches {cnt[k]})")
    if not acc: print(p, "no rows; tail of log:"); print(open(O + f"/{p}.log").read()[-1500:])
PY
