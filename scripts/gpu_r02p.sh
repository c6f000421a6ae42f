#!/bin/bash
# round 2, call P: node count (virtual ranks), config 3/4/5 single-GPU legs
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_db_device.py -m gpu -x -q -k "node" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "other_baseline" 2>&1 | tail -15
} > gpurun_out/r02p_tests.log 2>&1
tail -40 gpurun_out/r02p_tests.log
