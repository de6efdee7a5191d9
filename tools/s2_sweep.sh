#!/bin/bash
# Development aid: A/B of the strip-kernel variants on one box.  Usage: tools/s2_sweep.sh OUTDIR "ENV1=.. ENV2=.." "ENV..." ...
# Every argument after OUTDIR is one configuration (space-separated VAR=value pairs, "" = defaults).
out=$1; shift
mkdir -p "$out"
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "== $i: $cfg" >> "$out/sweep.log"
  env $cfg timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants > "$out/b$i.json" 2>> "$out/err.txt" || exit 1
  python - "$out/b$i.json" "$cfg" >> "$out/sweep.log" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
ks=[(r['kernel'],r['avg_launch_ms'])]+[(o['kernel'],o['avg_launch_ms']) for o in r['other_kernels']]
short={'fused':'cgstep','pcg_u':'update','prolo':'prolong','Jacob':'sweep','resid':'restrict','first':'first2'}
print(f"{d['ms_per_step']:8.3f} ms/step it {d['solver_iters']['fwd']}+{d['solver_iters']['adj']} err {d['parity_vs_oracle']['vs_exact_solution']['dkappa_rel_err_max']:.1e} | "+' '.join(f"{short[k[:5]]}={v:.4f}" for k,v in sorted(ks)))
PY
done
cat "$out/sweep.log"
