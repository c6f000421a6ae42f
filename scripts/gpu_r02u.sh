#!/bin/bash
mkdir -p gpurun_out
MGC_BENCH_FORCE_SHARDED=1 timeout 150 python bench.py --steps 2 --warmup 1 --no-e2e --no-cpu-baseline --no-check > gpurun_out/r02u_sharded.json 2> gpurun_out/r02u_sharded.err
echo "rc $?"; cut -c1-1800 gpurun_out/r02u_sharded.json; tail -3 gpurun_out/r02u_sharded.err
