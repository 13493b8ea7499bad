#!/bin/bash
# usage: tools/ab.sh <rounds> <variant suffixes...>   (interleaved rounds in one box)
R=$1; shift
for i in $(seq 1 $R); do for v in "$@"; do
  lib=$PWD/bayesian_dlms_amd/libdlm_engine$v.so
  DLM_ENGINE_LIB=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', round(j['value']/1e6,1), round(j['roofline']['forward_ms'],3), round(j['roofline']['backward_ms'],3))"
done; done
