#!/bin/bash
# On the GPU box: tools/kt_timeline.sh <name> <bench args...> -- rocprofv3 kernel trace of a short bench run; prints the kernels of the last step with start / end
# times relative to the step's first kernel and the queue they ran on
name=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$name; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary "$@" > $O/bench.json 2> $O/err.txt
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys, json
O = sys.argv[1]
rows = [r for r in csv.DictReader(open(glob.glob(O + "/kt/**/*kernel_trace.csv", recursive=True)[0])) if "dlm" in r["Kernel_Name"] and "simulate" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: kernels after the last gap of more than 0.3 ms before a count_gaps / first kernel ... simply take the tail of one step's worth
names = [r["Kernel_Name"] for r in rows]
first = names[0]
starts = [i for i, n in enumerate(names) if n == first]
lo = starts[-2] if len(starts) >= 2 else 0      # the second to last occurrence of the step's first kernel: a full timed step (the last one is the COUNT_STEPS call)
hi = starts[-1] if len(starts) >= 2 else len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} -> {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  q{r['Queue_Id']:>2} grid {r['Grid_Size_X']:>8} {r['Kernel_Name'].replace('void dlm::', '')[:70]}")
try:
    b = json.load(open(O + "/bench.json")); print("bench ms_per_step", b["ms_per_step"], b["roofline"]["forward_ms"], b["roofline"]["backward_ms"])
except Exception as e:
    print("no bench line", e)
PY
