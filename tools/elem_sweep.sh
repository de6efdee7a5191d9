#!/bin/bash
out=${1:-gpurun_out/elemsweep}; mkdir -p $out
run() {  # label, env...
  label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --only-variant kappa_element_field --variant-reps 3 --steps 2 --warmup 1 > $out/$label.json 2> $out/$label.err || { echo "$label FAILED"; tail -3 $out/$label.err; return 1; }
  python - $out/$label.json $label <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
v = d["variants"]["kappa_element_field"]
print(f"{sys.argv[2]:20s} {v['ms_per_step']:8.2f} ms  {v['value_per_gpu']:7.1f} solves/s  its {v['iters_fwd']}+{v['iters_adj']}  all {v['ms_all']}", flush=True)
PY
}
run ps2 DIFFHE_FUSED_PS=2
run ps3 DIFFHE_FUSED_PS=3
