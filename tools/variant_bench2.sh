#!/bin/bash
# like variant_bench.sh, with the per-element variant reported as well
knob=$1; shift
for v in "$@"; do
  env $knob=$v python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$knob=$v', 'value', d['value'], 'ms/step', d['ms_per_step'], 'per-element field', d['variants']['kappa_element_field']['value_per_gpu'], 'fp64 V-cycle', d['variants']['vcycle_storage_fp64']['value_per_gpu'])" || exit 1
done
