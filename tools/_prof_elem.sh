set -e -o pipefail
cd /root/repo
mkdir -p gpurun_out/r2u
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2u/prof -o elem -- python3 bench.py --kappa element --steps 2 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r2u/bench_elem.log 2>&1
tail -1 gpurun_out/r2u/bench_elem.log | cut -c1-600
find gpurun_out/r2u/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2u/elem_kernel_stats.csv
python tools/roofline_table.py $(find gpurun_out/r2u/prof -name "*kernel_trace.csv" | head -1) > gpurun_out/r2u/elem_roofline.md; find gpurun_out/r2u/prof -name "*kernel_trace.csv" | xargs rm -f
head -30 gpurun_out/r2u/elem_kernel_stats.csv | cut -c1-220
