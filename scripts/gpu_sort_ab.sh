#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "radix or golden or session" > gpurun_out/pytest_sort.log 2>&1
echo "pytest exit $?"; tail -5 gpurun_out/pytest_sort.log
timeout 600 python scripts/sort_bench.py ${SORT_N:-135000000} 36 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sort_ab.log
