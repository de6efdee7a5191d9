#!/bin/bash
out=${1:-gpurun_out/trsweep3}; mkdir -p $out
run() {  # label, env...
  label=$1; shift
  env "$@" timeout -k 10 120 python bench.py --no-variants --no-cpu-baseline --steps 8 --warmup 2 --kernel-reps 1 > $out/$label.json 2> $out/$label.err || { echo "$label FAILED"; tail -3 $out/$label.err; return 1; }
  python - $out/$label.json $label <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
o = {k["symbol"].split("<")[0]: k["avg_launch_ms"] for k in r["other_kernels"]}
print(f"{sys.argv[2]:20s} step {d['ms_per_step']:8.3f} ms  median {d['timed_steps_ms']['median']:8.3f} upd {r['avg_launch_ms'] if 'update' in r['kernel'] else [k['avg_launch_ms'] for k in r['other_kernels'] if 'update' in k['kernel']]} all {o} iters {d['solver_iters']['fwd']}+{d['solver_iters']['adj']} parity {d['parity_vs_oracle']['vs_exact_solution']['u_rel_err_max']:.2e} {d['parity_vs_oracle']['vs_exact_solution']['dkappa_rel_err_max']:.2e}", flush=True)
PY
}
run cw2 X=1
run cw1 DIFFHE_PRE_CW=1
run cw1_tr8 DIFFHE_PRE_CW=1 DIFFHE_FUSED_TR_PRE=8
run cw1_tr16 DIFFHE_PRE_CW=1 DIFFHE_FUSED_TR_PRE=16
run cw2b X=1
