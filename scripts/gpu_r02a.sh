#!/bin/bash
# round 2, call A: new tests first (device encoder, sharded db, digests, full size), then the whole suite, smoke, default bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2; df -h /dev/shm | tail -1) > $OUT/env.log 2>&1
echo "== new tests"
timeout 900 python -m pytest tests/test_db_device.py -q -p no:cacheprovider --maxfail=30 > $OUT/pytest_new.log 2>&1
echo "new tests exit $?"; tail -40 $OUT/pytest_new.log
echo "== digests + full size"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "digests or config1_full" --durations=5 > $OUT/pytest_full.log 2>&1
echo "full exit $?"; tail -25 $OUT/pytest_full.log
echo "== rest of the gpu suite"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 --deselect tests/test_gpu_parity.py::test_config1_full_size_matches_threaded_port --ignore=tests/test_db_device.py > $OUT/pytest_gpu.log 2>&1
echo "suite exit $?"; tail -15 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
echo "== default bench"
timeout 1500 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench exit $?"; cat $OUT/bench_full.json; tail -8 $OUT/bench_full.err
