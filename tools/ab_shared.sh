#!/bin/bash
# On the GPU box: shared-covariance path against the per-series kernels (DLM_OPT_SHARED_COV = 16777216 turns it on), interleaved, several batch sizes
for rep in 1 2; do
for n in 10000 5000 2500 1250; do
  a=$(python bench.py --steps 15 --warmup 3 --series $n --no-cpu-baseline --flags 16777216 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])")
  b=$(python bench.py --steps 15 --warmup 3 --series $n --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])")
  echo "series $n: shared $a ms   per-series $b ms"
done
done
