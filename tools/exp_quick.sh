#!/bin/bash
# quick look: headline step + in-solver kernel times (+ optional per-element-field variant): bash tools/exp_quick.sh tag [elem]
cd $GRAFT_REPO_ROOT
tag=${1:-q}
A="--steps 5 --warmup 2 --no-cpu-baseline"
if [ "$2" == "elem" ]; then V="--only-variant kappa_element_field"; else V="--no-variants"; fi
python bench.py $A $V > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err || { tail -5 gpurun_out/${tag}.err; exit 1; }
python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/${tag}.json") if l.startswith("{")][0])
r=j["roofline"]
print("${tag} ms/step", j["ms_per_step"], "iters", j["solver_iters"]["fwd"], j["solver_iters"]["adj"], "step frac", r["step"]["frac"])
for row in [r]+r["other_kernels"]:
    print("   %.4f ms  %s" % (row["avg_launch_ms"], row["kernel"][:70]))
v=(j.get("variants") or {}).get("kappa_element_field")
if v: print("   kappa_element_field", v["value_per_gpu"], "solves/s", v["ms_per_step"], "ms", v["iters_fwd"], v["iters_adj"])
PY
