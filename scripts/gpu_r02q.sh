#!/bin/bash
# round 2, call Q: narrowed grouping passes -- parity, then the bench
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "narrowed or oversized or repeat_family or other_baseline" 2>&1 | tail -15
timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e 2>&1 | tail -3
MGC_NARROW=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e --no-check 2>&1 | tail -1
} > gpurun_out/r02q.log 2>&1
tail -30 gpurun_out/r02q.log
