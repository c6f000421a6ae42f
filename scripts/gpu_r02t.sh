#!/bin/bash
# round 2, call T: fifteen-bit file histogram + high digit first
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "narrowed or oversized or repeat_family or other_baseline or medium_scale" 2>&1 | tail -15
timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline 2>&1 | tail -1
MGC_FINE_HIST=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e --no-check --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1200
} > gpurun_out/r02t.log 2>&1
tail -30 gpurun_out/r02t.log
