#!/bin/bash
# Kernel stats of the per-element-kappa variant (one matrix per sample and level); run on the GPU box from the repo root.
out=${1:-gpurun_out/prof_elem}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -- python3 $R/bench.py --kappa element --steps 2 --warmup 1 --no-cpu-baseline --no-variants > $R/$out/bench_element_under_rocprof.json 2> $R/$out/trace.err || exit 1
cd $R
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kappa_element_kernel_stats.csv
python tools/gap_report.py $(find $out/trace -name "*kernel_trace.csv" | head -1) 5 > $out/element_gap_report.txt
rm -rf $out/trace
head -30 $out/kappa_element_kernel_stats.csv | cut -c1-150
