#!/bin/bash
# every bench configuration once (run on the GPU box): tools/bench_all.sh <outdir>
out=${1:-gpurun_out/bench_all}; mkdir -p $out
set -e
python bench.py > $out/bench_c2.json 2> $out/bench_c2.err
python bench.py --records packed --no-cpu-baseline > $out/bench_c2_packed.json 2> $out/bench_c2_packed.err
python bench.py --missing 0.05 --no-cpu-baseline > $out/bench_c2_missing.json 2> $out/bench_c2_missing.err
python bench.py --semantics literal-q1 --no-cpu-baseline --steps 3 --warmup 1 > $out/bench_c2_literal.json 2> $out/bench_c2_literal.err
python bench.py --config c3 > $out/bench_c3.json 2> $out/bench_c3.err
python bench.py --config c3 --sampler simsmooth --no-cpu-baseline > $out/bench_c3_sim.json 2> $out/bench_c3_sim.err
python bench.py --config c4 > $out/bench_c4.json 2> $out/bench_c4.err
python bench.py --config c5 > $out/bench_c5.json 2> $out/bench_c5.err
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r = j["roofline"]
print(sys.argv[1].split("/")[-1], "%.4g" % j["value"], "%.3f ms" % j["ms_per_step"], r["kernel"], "frac %.3f" % r["frac"], "fwd %.3f bwd %.3f" % (r["forward_ms"], r["backward_ms"]),
      "copy %.0f GB/s" % r["peak_measured"] if "peak_measured" in r else "", "cpu %.3g" % j["cpu_baseline"]["value"] if "cpu_baseline" in j else "")
PY
done
