#!/bin/bash
# Same-box sweep of the fused passes' tile heights (rows per tile, per multigrid level).
out=${1:-gpurun_out/trsweep}; mkdir -p $out
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
for t in 27 34 45 69 147; do run post0_$t DIFFHE_FUSED_TR_POST=$t; done
for t in 13 17 23 35 74; do run pre0_$t DIFFHE_FUSED_TR_PRE=$t; done
for t in 16 24 35 52; do run post1_$t DIFFHE_FUSED_TR_POST=0:$t; done
for t in 8 12 18 26; do run pre1_$t DIFFHE_FUSED_TR_PRE=0:$t; done
for t in 16 32 64; do run post2_$t DIFFHE_FUSED_TR_POST=0:0:$t; done
for t in 8 16 32; do run pre2_$t DIFFHE_FUSED_TR_PRE=0:0:$t; done
run base2 X=1
