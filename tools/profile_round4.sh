#!/bin/bash
# On the GPU box: for every bench configuration one rocprofv3 --kernel-trace --stats pass and the two HBM counter passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, TCC counters), each in its own run without any other
# trace domain.  Summaries go to gpurun_out/prof_r04; copy what is to be judged into profiles/.
#   tools/profile_round4.sh [config-filter-regex]
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FILTER=${1:-.}
CONFIGS=("c2" "c2 --flags 268435456" "c2 --flags 4194304" "c2 --missing 0.05" "c2 --semantics literal-q1" "c2 --flags 268435456 --semantics literal-q1" "c3" "c3 --flags 67108864" "c3 --series 1250" "c3 --sampler simsmooth" "c4" "c4 --flags 4194304" "c4g" "c4g --flags 67108864" "c5")
for c in "${CONFIGS[@]}"; do
  tag=$(echo $c | tr -d ' -.')
  if ! echo "$tag" | grep -Eq "$FILTER"; then continue; fi
  echo "== $c" >&2
  ST="--steps 3 --warmup 1"; if [ "$tag" = "c2" ]; then ST="--steps 10 --warmup 3"; fi   # (the headline: the protocol of the driver-visible line)
  rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o run --output-format csv -- python3 $R/bench.py --config $c $ST --no-cpu-baseline --no-secondary > $O/bench_${tag}_under_rocprof.json 2> $O/kt_$tag.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_$tag -o run --output-format csv -- python3 $R/bench.py --config $c --steps 2 --warmup 0 --no-cpu-baseline --no-secondary > $O/fetch_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_$tag -o run --output-format csv -- python3 $R/bench.py --config $c --steps 2 --warmup 0 --no-cpu-baseline --no-secondary > $O/write_$tag.log 2>&1
done
cd $R
python3 - "$O" <<'PY'
import csv, glob, hashlib, json, os, re, sys, collections
O = sys.argv[1]
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
def source_of(k):
    if "svd" in k: return "dlm_svd.hip"
    if "w48" in k: return "dlm_wave48.hip"
    if any(s in k for s in ("k_sampler_sp16", "k_mean_sampler_sp16", "rts16", "k_rts_decide", "k_normals4", "k_mark_gaps")): return "dlm_sampler16.hip"
    if "sp16" in k or "k_count_gaps" in k: return "dlm_sparse16.hip"
    if "tiled" in k: return "dlm_tiled.hip"
    if "generic" in k or "stats_pool" in k: return "dlm_generic.hip"
    if "lane" in k: return "dlm_lane.hip"
    return None
def sha16(f):
    return hashlib.sha256(open(os.path.join(ROOT, "bayesian_dlms_amd", "csrc", f), "rb").read()).hexdigest()[:16]
def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(dlm::KArgs.*", "", name)
    return name.replace("dlm::w48::", "").replace("dlm::s16::", "").replace("dlm::", "")
for d in glob.glob(O + "/kt_*"):
    st = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if st:
        rows = list(csv.reader(open(st[0])))
        keep = [rows[0]] + [r for r in rows[1:] if r and "dlm" in r[0]]
        csv.writer(open(O + "/r04_kernel_stats_" + d.split("kt_")[-1] + ".csv", "w")).writerows(keep)
out = {"note": ("rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes of `python3 bench.py --config <c> --steps 2 --warmup 0 --no-cpu-baseline --no-secondary` "
                "(tools/profile_round4.sh); per-launch averages per (kernel, grid size).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the "
                "bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM): read bytes = 2 * FETCH_SIZE * 1024.  source_sha16: sha256[:16] of the .hip file that "
                "defines the kernel when the counters were taken -- bench.py drops a figure whose source has changed since."),
       "configs": {}}
tags = sorted({os.path.basename(p)[len("fetch_"):] for p in glob.glob(O + "/fetch_*") if os.path.isdir(p)})
for tag in tags:
    per = {}
    for name, key in (("fetch", "FETCH_SIZE_KiB"), ("write", "WRITE_SIZE_KiB")):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for fn in glob.glob(O + f"/{name}_{tag}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                if "dlm" not in r["Kernel_Name"]: continue
                k = (short(r["Kernel_Name"]), int(r.get("Grid_Size", 0) or 0))
                acc[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k in acc:
            per.setdefault(k, {})[key] = acc[k] / cnt[k]
            per[k]["launches_" + name] = cnt[k]
    ents = []
    for (k, grid), v in sorted(per.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE_KiB", 0) * 2 + kv[1].get("WRITE_SIZE_KiB", 0))):
        f, w = v.get("FETCH_SIZE_KiB"), v.get("WRITE_SIZE_KiB")
        if f is None or w is None: continue
        src = source_of(k)
        ents.append({"kernel": k, "grid_threads": grid, "launches": v.get("launches_fetch"), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                     "hbm_read_bytes": 2 * f * 1024, "hbm_write_bytes": w * 1024, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
                     "source": src, "source_sha16": sha16(src) if src else None})
    out["configs"][tag] = ents
json.dump(out, open(O + "/r04_hbm_traffic.json", "w"), indent=1)
for tag, ents in out["configs"].items():
    print(tag)
    for e in ents[:6]:
        print(f"   {e['kernel'][:64]:64s} grid {e['grid_threads']:>8} x{e['launches']}: read {e['hbm_read_bytes'] / 1e9:8.3f} GB, written {e['hbm_write_bytes'] / 1e9:8.3f} GB")
PY
