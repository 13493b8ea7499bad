#!/bin/bash
# On the GPU box: tools/ab_libs.sh <variant-name> <bench args...> -- the same bench run with the regular library and with bayesian_dlms_amd/libdlm_engine_<variant>.so, twice each, alternating
v=$1; shift
for i in 1 2; do
  for lib in "" $v; do
    if [ -n "$lib" ]; then export DLM_ENGINE_LIB=$GRAFT_REPO_ROOT/bayesian_dlms_amd/libdlm_engine_$lib.so; else unset DLM_ENGINE_LIB; fi
    python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-regular}', round(j['ms_per_step'],3), round(j['roofline']['forward_ms'],3), round(j['roofline']['backward_ms'],3))"
  done
done
