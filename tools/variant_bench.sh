#!/bin/bash
# Development aid: run bench.py under several values of an environment knob and print value / dominant-kernel time.
#   tools/variant_bench.sh DIFFHE_PUPD_VARIANT 0 1 2 3
knob=$1; shift
for v in "$@"; do
  env $knob=$v python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$knob=$v', 'value', d['value'], 'ms/step', d['ms_per_step'], 'fused-step ms (in solver)', r['avg_launch_ms'], 'isolated', r['isolated_launch_ms'])" || exit 1
done
