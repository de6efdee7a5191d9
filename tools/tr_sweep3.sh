#!/bin/bash
# Same-box A/B of environment knobs on the headline workload (short bench runs, per-kernel in-solver times):
#   tools/tr_sweep3.sh <outdir> <label> "<ENV=.. ENV=..>" [<label> "<env>" ...]      ("" or X=1 for the defaults)
# NOTE (round 3, fp32-pair residual): an A/B inside ONE binary can flatter a new code path when its mere presence slows the
# old path of the same kernel -- confirm against the previous binary's kernel times before keeping anything.
out=${1:-gpurun_out/ab}; mkdir -p $out; shift
run() {  # label, env...
  label=$1; shift
  env "$@" timeout -k 10 120 python bench.py --no-variants --no-cpu-baseline --steps 8 --warmup 2 --kernel-reps 1 > $out/$label.json 2> $out/$label.err || { echo "$label FAILED"; tail -3 $out/$label.err; return 1; }
  python - $out/$label.json $label <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
o = {k["kernel"][:22]: k["avg_launch_ms"] for k in [r] + r["other_kernels"]}
print(f"{sys.argv[2]:20s} step {d['ms_per_step']:8.3f} ms  median {d['timed_steps_ms']['median']:8.3f}  {o}  iters {d['solver_iters']['fwd']}+{d['solver_iters']['adj']} parity {d['parity_vs_oracle']['vs_exact_solution']['u_rel_err_max']:.2e} {d['parity_vs_oracle']['vs_exact_solution']['dkappa_rel_err_max']:.2e}", flush=True)
PY
}
if [ $# -eq 0 ]; then set -- base X=1 base2 X=1; fi
while [ $# -ge 2 ]; do
  lab=$1; envs=${2:-X=1}; shift 2
  run $lab $envs
done
