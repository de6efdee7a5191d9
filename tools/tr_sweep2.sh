#!/bin/bash
out=${1:-gpurun_out/trsweep2}; mkdir -p $out
run() {  # label, env...
  label=$1; shift
  env "$@" timeout -k 10 120 python bench.py --no-variants --no-cpu-baseline --steps 8 --warmup 2 --kernel-reps 1 > $out/$label.json 2> $out/$label.err || { echo "$label FAILED"; return 1; }
  python - $out/$label.json $label <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
o = {k["symbol"].split("<")[0]: k["avg_launch_ms"] for k in d["roofline"]["other_kernels"]}
print(f"{sys.argv[2]:28s} step {d['ms_per_step']:8.3f} ms  median {d['timed_steps_ms']['median']:8.3f}  pre {o.get('fused_pre_kernel')}  post {o.get('fused_post_kernel')}  iters {d['solver_iters']['fwd']}+{d['solver_iters']['adj']}", flush=True)
PY
}
run base X=1
run spl1 DIFFHE_FUSED_SPL=1
for t in 16 22 30 64; do run spl1_post0_$t DIFFHE_FUSED_SPL=1 DIFFHE_FUSED_TR_POST=$t; done
for t in 8 11 16 32; do run spl1_pre0_$t DIFFHE_FUSED_SPL=1 DIFFHE_FUSED_TR_PRE=$t; done
for t in 12 16 18; do run post0_$t DIFFHE_FUSED_TR_POST=$t; done
for t in 6 8 9; do run pre0_$t DIFFHE_FUSED_TR_PRE=$t; done
