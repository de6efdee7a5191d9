#!/bin/bash
# Only the two PMC passes of tools/profile_round.sh (default bench workload); GPU box, repo root.
out=${1:-gpurun_out/pmc}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P="--steps 1 --warmup 1 --no-cpu-baseline --no-variants --kernel-reps 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/fetch -- python3 $R/bench.py $P > $R/$out/bench_fetch.json 2> $R/$out/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/write -- python3 $R/bench.py $P > $R/$out/bench_write.json 2> $R/$out/write.err || exit 1
cd $R
python tools/pmc_reduce.py $out/fetch $out/write $out/pmc_traffic.json > $out/pmc_reduce.log || exit 1
rm -rf $out/fetch $out/write
ls -la $out
