#!/bin/bash
# block width of the fused passes (DIFFHE_FUSED_NW = waves per block): step time, in-solver kernel times, PMC traffic
cd $GRAFT_REPO_ROOT
A="--steps 5 --warmup 2 --no-cpu-baseline --no-variants"
for nw in 4 8 16; do
  export DIFFHE_FUSED_NW=$nw
  python bench.py $A > gpurun_out/r4c_nw$nw.json 2> gpurun_out/r4c_nw$nw.err || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4c_nw$nw.json") if l.startswith("{")][0])
r=j["roofline"]
print("NW=$nw ms/step", j["ms_per_step"], "iters", j["solver_iters"]["fwd"], j["solver_iters"]["adj"])
for row in [r]+r["other_kernels"]:
    print("   %.4f ms  %s" % (row["avg_launch_ms"], row["kernel"][:70]))
PY
done
for nw in 8 16; do
  export DIFFHE_FUSED_NW=$nw
  bash tools/pmc_round.sh gpurun_out/r4c_pmc_nw$nw > /dev/null 2>&1 || exit 1
  python - <<PY
import json
j=json.load(open("gpurun_out/r4c_pmc_nw$nw/pmc_traffic.json"))
for k,v in j["kernels"].items():
    if "fused" in k: print("NW=$nw  %-75s rd %.3f wr %.3f tot %.3f" % (k[:75], v["read_passes"], v["write_passes"], v["passes"]))
PY
done
