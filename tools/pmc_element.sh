#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the per-element-kappa variant; GPU box, repo root.
out=${1:-gpurun_out/pmc_elem}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P="--kappa element --steps 1 --warmup 1 --no-cpu-baseline --no-variants --kernel-reps 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/fetch -- python3 $R/bench.py $P > $R/$out/bench_fetch.json 2> $R/$out/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/write -- python3 $R/bench.py $P > $R/$out/bench_write.json 2> $R/$out/write.err || exit 1
cd $R
python tools/pmc_reduce.py $out/fetch $out/write $out/pmc_traffic_element.json > /dev/null
rm -rf $out/fetch $out/write
python - $out/pmc_traffic_element.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    print(f"{k[:88]:88s} passes {v['passes']:.3f} (R {v['read_passes']:.3f} W {v['write_passes']:.3f})")
PY
