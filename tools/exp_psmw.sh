#!/bin/bash
# per-sample kappa fields: fused passes at 2 or 3 waves per SIMD (DIFFHE_FUSED_PS_MW bit 0 = PRE, bit 1 = POST); config 5 shard
cd $GRAFT_REPO_ROOT
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r4e_counters.txt 2>&1)
A="--steps 2 --warmup 1 --no-cpu-baseline"
for mw in 0 2 3; do
  export DIFFHE_FUSED_PS_MW=$mw
  python bench.py $A --only-variant kappa_element_field > gpurun_out/r4e_mw$mw.json 2> gpurun_out/r4e_mw$mw.err || { tail -5 gpurun_out/r4e_mw$mw.err; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4e_mw$mw.json") if l.startswith("{")][0])
v=j["variants"]["kappa_element_field"]
print("PS_MW=$mw", v["value_per_gpu"], "solves/s", v["ms_per_step"], "ms", v["ms_all"], "iters", v["iters_fwd"], v["iters_adj"], "headline", j["ms_per_step"])
PY
done
unset DIFFHE_FUSED_PS_MW
python bench.py $A --only-variant config5_512_b64_adam > gpurun_out/r4e_c5.json 2> gpurun_out/r4e_c5.err || { tail -5 gpurun_out/r4e_c5.err; exit 1; }
python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4e_c5.json") if l.startswith("{")][0])
v=j["variants"]["config5_512_b64_adam"]
print("config5", {k: v[k] for k in ("adam_steps_per_s","solves_per_s","ms_per_adam_step","ms_all","iterations")})
print("   roofline", v["roofline"]["kernel"][:60], v["roofline"]["avg_launch_ms"], v["roofline"]["frac"])
PY
