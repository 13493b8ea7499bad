#!/bin/bash
# On the GPU box: tools/kt.sh <name> <bench args...> -- rocprofv3 kernel-trace stats of one bench run, summary printed and kept under gpurun_out/<name>
name=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$name; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/bench.json 2> $O/kt.err
cd $R
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"].split("(")[0][:70]:72s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} {r["Percentage"]:>6s}%')
PY
cut -c1-250 $O/bench.json
