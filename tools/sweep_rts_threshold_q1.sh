#!/bin/bash
# On the GPU box: the literal-Q1 call through the shared RTS tables (DLM_OPT_NO_SMALL_BATCH = 2097152) against k_smoother_rts16 per series (DLM_OPT_SMOOTHER_PER_SERIES =
# 268435456), by batch size: where the tables start to pay (DLM_RTS_SHARED_MIN).
O=gpurun_out/r04/thrq1; mkdir -p $O
for n in 64 256 512 1024 2048 4096; do
  for f in 2097152 268435456; do
    python bench.py --semantics literal-q1 --series $n --flags $f --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/n${n}_f$f.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob, re
rows = {}
for fn in glob.glob("gpurun_out/r04/thrq1/n*_f*.json"):
    n, f = map(int, re.findall(r"n(\d+)_f(\d+)", fn)[0])
    j = json.loads(open(fn).read().strip().splitlines()[-1])
    rows.setdefault(n, {})[f] = (j["ms_per_step"], j["config"]["variant"])
out = []
for n in sorted(rows):
    a, b = rows[n].get(2097152), rows[n].get(268435456)
    print(f"{n:6d} series: tables {a[0]:7.3f} ms ({a[1]})   per-series {b[0]:7.3f} ms ({b[1]})")
    out.append({"series": n, "tables_ms": a[0], "per_series_ms": b[0]})
json.dump(out, open("gpurun_out/r04/thrq1/r04_rts_threshold_q1.json", "w"), indent=1)
PY
