#!/bin/bash
# same-box A/B of the default bench (fp32 fused rollout, 65 536 envs): tools/ab_run.sh <other.so> [bench args]
other=${1:?other .so}; shift
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for lib in "" "$other"; do
    RAPTOR_QUAD_LIB=$lib python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('${lib:-working tree}', d['ms_per_step'] * 1e3, 'us/step', d['value'])"
  done
done
