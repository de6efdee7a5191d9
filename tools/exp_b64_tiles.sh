#!/bin/bash
# 512^2 x 64 (config 5 shard shape): tile heights of the fused passes against the resident-block count
#   bash tools/exp_b64_tiles.sh "pre:post" "pre:post" ...     (0 = default)
cd $GRAFT_REPO_ROOT
A="--mesh 512 --batch 64 --steps 20 --warmup 3 --no-cpu-baseline --no-variants"
for pp in "$@"; do
  pre=${pp%%:*}; post=${pp##*:}
  unset DIFFHE_FUSED_TR_PRE DIFFHE_FUSED_TR_POST
  [ "$pre" != "0" ] && export DIFFHE_FUSED_TR_PRE=$pre
  [ "$post" != "0" ] && export DIFFHE_FUSED_TR_POST=$post
  python bench.py $A > gpurun_out/b64t_${pre}_${post}.json 2> gpurun_out/b64t_${pre}_${post}.err || { tail -5 gpurun_out/b64t_${pre}_${post}.err; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/b64t_${pre}_${post}.json") if l.startswith("{")][0])
r=j.get("roofline")
ks=" ".join("%.1f" % (1e3*row["avg_launch_ms"]) for row in [r]+r["other_kernels"]) if r else ""
print("pre=$pre post=$post ms/step", j["ms_per_step"], "iters", j["solver_iters"]["fwd"], j["solver_iters"]["adj"], "| kernel us:", ks)
PY
done
