#!/bin/bash
# config-5 shape (512^2 x 64 per GPU): where a step goes, new (one-sample fused) vs old (single passes) kernels
cd $GRAFT_REPO_ROOT
A="--mesh 512 --batch 64 --steps 20 --warmup 3 --no-cpu-baseline --no-variants"
for s2 in 1 0; do
  export DIFFHE_STRIP2=$s2
  python bench.py $A > gpurun_out/r4f_s$s2.json 2> gpurun_out/r4f_s$s2.err || { tail -5 gpurun_out/r4f_s$s2.err; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4f_s$s2.json") if l.startswith("{")][0])
print("STRIP2=$s2 ms/step", j["ms_per_step"], "value", j["value"], "iters", j["solver_iters"]["fwd"], j["solver_iters"]["adj"], "launches/step", j["roofline"]["step"]["launches_per_step"] if j.get("roofline") else None)
r=j.get("roofline")
if r:
    for row in [r]+r["other_kernels"]:
        print("   %.4f ms x %d  %s" % (row["avg_launch_ms"], row["launches_timed"], row["kernel"][:70]))
PY
done
export DIFFHE_STRIP2=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4f_trace -- python3 $R/bench.py --mesh 512 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline --no-variants > $R/gpurun_out/r4f_prof.json 2> $R/gpurun_out/r4f_prof.err
cd $R
st=$(find gpurun_out/r4f_trace -name "*kernel_stats.csv" | head -1); cp $st gpurun_out/r4f_kernel_stats.csv
rm -rf gpurun_out/r4f_trace
head -40 gpurun_out/r4f_kernel_stats.csv | cut -c1-200
