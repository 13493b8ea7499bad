#!/bin/bash
# usage: tools/sweep_small.sh <lib suffixes...>   per-GPU share of the strong-scaling contract on one GPU (10 000 series over G GPUs)
for v in "$@"; do for n in 10000 5000 2500 1250 625; do
  lib=$PWD/bayesian_dlms_amd/libdlm_engine$v.so
  DLM_ENGINE_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --series $n 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', $n, round(j['ms_per_step'],3), round(j['roofline']['forward_ms'],3), round(j['roofline']['backward_ms'],3))"
done; done
