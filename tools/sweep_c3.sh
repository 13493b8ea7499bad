#!/bin/bash
# On the GPU box: C3 per Gibbs iteration against the batch size, shared factors (forced) and per-series sampler
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
for n in 320 640 1250 2500 5000 10000; do
  for f in 2097152 67108864; do
    python3 $R/bench.py --config c3 --series $n --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --flags $f > $O/c3_${n}_$f.json 2>> $O/err.txt
    python3 - $O/c3_${n}_$f.json $n $f <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "shared" if sys.argv[3] == "2097152" else "per-series", j["config"]["variant"], "ms/iter %.3f" % j["ms_per_step"], "fwd %.3f bwd %.3f" % (j["roofline"]["forward_ms"], j["roofline"]["backward_ms"]))
PY
  done
done
