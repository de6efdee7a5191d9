#!/bin/bash
out=${1:-gpurun_out/gaps}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/$out/trace -- python3 $R/bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-variants --kernel-reps 1 $EXTRA > $R/$out/bench.json 2> $R/$out/trace.err || exit 1
cd $R
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/gap_report.py $kt 40 > $out/gap_report.txt
rm -rf $out/trace
cat $out/gap_report.txt
