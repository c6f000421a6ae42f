#!/bin/bash
# round 2, call X: per-phase cycle stamps of the two narrowed grouping passes (first two files of the judged workload)
mkdir -p gpurun_out
MGC_GROUP_DBG=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-e2e --no-cpu-baseline --no-check 2>&1 | grep groupdbg > gpurun_out/r02x_groupdbg.log
cat gpurun_out/r02x_groupdbg.log
