#!/bin/bash
# usage: tools/ab_flags.sh <rounds> <series> <flags...>   (interleaved rounds in one box, the shipped library, bench.py --flags F)
R=$1; N=$2; shift; shift
for i in $(seq 1 $R); do for f in "$@"; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --series $N --flags $f 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('flags $f N $N', round(j['value']/1e6,1), round(j['ms_per_step'],3), round(j['roofline']['forward_ms'],3), round(j['roofline']['backward_ms'],3))"
done; done
