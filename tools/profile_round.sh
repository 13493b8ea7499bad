#!/bin/bash
# On the GPU box: kernel-trace stats + the two HBM PMC passes of the default bench; copies summaries to gpurun_out/prof_final
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o run --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
python3 bench.py --steps 10 --warmup 3 > $O/bench_plain.json 2> $O/bench_plain.err
python3 - <<PY
import csv, glob, json, collections
O = "$O"
st = glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True)
if st:
    open(O + "/kernel_stats.csv", "w").write(open(st[0]).read())
res = {}
for name in ("fetch", "write"):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for fn in glob.glob(O + f"/pmc_{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            k = "k_filter_sp16" if "k_filter_sp16" in k else "k_smoother_sp16" if "k_smoother_sp16" in k else None
            if k: acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    res[name] = {k: acc[k] / cnt[k] for k in acc}
json.dump(res, open(O + "/pmc_raw.json", "w"), indent=1)
print(res)
PY
