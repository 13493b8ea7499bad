#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace stats of every bench configuration + the two HBM PMC passes of the headline;
# writes summaries to gpurun_out/prof_r03 (copy what is to be judged into profiles/).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o run --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_c2_under_rocprof.json 2> $O/kt_c2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary > $O/pmc_write.log 2>&1
for c in "c2 --flags 4194304" "c2 --missing 0.05" "c2 --semantics literal-q1" "c2 --records packed" "c2 --flags 16777216" "c3" "c3 --flags 67108864" "c3 --series 1250" "c3 --sampler simsmooth" "c4" "c4 --flags 4194304" "c4g" "c4g --flags 67108864" "c5"; do
  tag=$(echo $c | tr -d ' -.')
  rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o run --output-format csv -- python3 $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_${tag}_under_rocprof.json 2> $O/kt_$tag.err
done
cd $R
python3 - <<PY
import csv, glob, json, collections
O = "$O"
for d in glob.glob(O + "/kt_*"):
    st = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if st:
        rows = list(csv.reader(open(st[0])))
        keep = [rows[0]] + [r for r in rows[1:] if r and ("dlm::" in r[0] or "dlm" in r[0])]
        csv.writer(open(O + "/kernel_stats_" + d.split("kt_")[-1] + ".csv", "w")).writerows(keep)
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
N, T, rec = 10000, 1000, 182
out = {"note": "rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate passes, tools/profile_round3.sh) on python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline, round 3 kernels; per-launch averages.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM), so read bytes = 2 * FETCH_SIZE * 1024.",
       "kernels": {}}
for k, alg in (("k_filter_sp16", (8 + 8 * rec) * N * T), ("k_smoother_sp16", 16 * rec * N * T)):
    f, w = res["fetch"].get(k), res["write"].get(k)
    if f is None or w is None: continue
    out["kernels"][k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_read_bytes": 2 * f * 1024, "hbm_write_bytes": w * 1024,
                         "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024, "algorithmic_bytes_per_launch": float(alg)}
json.dump(out, open(O + "/r03_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
