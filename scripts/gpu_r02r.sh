#!/bin/bash
# round 2, call R: kernel stats of the narrowed passes
mkdir -p gpurun_out/r02r
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02r/prof -o full -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-check > gpurun_out/r02r/bench.json 2> gpurun_out/r02r/bench.err
cp $(find gpurun_out/r02r/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r02r/kernel_stats.csv
rm -rf gpurun_out/r02r/prof
cut -c1-180 gpurun_out/r02r/kernel_stats.csv | head -24
