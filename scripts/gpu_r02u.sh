#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_digit_buckets" 2>&1 | tail -12
MGC_BENCH_FORCE_SHARDED=1 MGC_SHARD_BITS=9 timeout 600 python bench.py --steps 2 --warmup 1 --no-e2e --no-cpu-baseline --no-check 2>&1 | tail -1 | cut -c1-1500
} > gpurun_out/r02u.log 2>&1
tail -16 gpurun_out/r02u.log
