#!/bin/bash
# Same-box check of the timed region's step-to-step spread: three default headline runs, per-step times only.
out=${1:-gpurun_out/jitter}; mkdir -p $out
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --no-variants --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err || exit 1
  python - $out/run$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["timed_steps_ms"])
PY
done
