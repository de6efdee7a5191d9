#!/bin/bash
# Kernel trace + the two PMC passes of the default bench workload (run on the GPU box, from the repo root).
#   tools/profile_round.sh <outdir under gpurun_out>
out=${1:-gpurun_out/prof}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--steps 3 --warmup 1 --no-cpu-baseline --no-variants"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -- python3 $R/bench.py $A > $R/$out/bench_under_rocprof.json 2> $R/$out/trace.err || exit 1
P="--steps 1 --warmup 1 --no-cpu-baseline --no-variants --kernel-reps 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/fetch -- python3 $R/bench.py $P > $R/$out/bench_fetch.json 2> $R/$out/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/write -- python3 $R/bench.py $P > $R/$out/bench_write.json 2> $R/$out/write.err || exit 1
cd $R
st=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp $st $out/kernel_stats.csv
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/roofline_table.py $kt > $out/roofline_table.md
python tools/gap_report.py $kt 40 > $out/step_budget.txt
python tools/pmc_reduce.py $out/fetch $out/write $out/pmc_traffic.json > $out/pmc_reduce.log
# the traces themselves are too large to bring back
rm -rf $out/trace $out/fetch $out/write
ls -la $out
