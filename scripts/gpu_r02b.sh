#!/bin/bash
# round 2, call B: staging/out-of-core/merge/union-sum/CLI tests, the suite (without the 3-minute full-size test), default bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp
echo "== new tests"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_cli.py tests/test_db_device.py -m gpu -q -p no:cacheprovider --maxfail=30 \
  -k "merge or out_of_core or text_file or cli or db_device or device_encoder or device_stream or session_database or forced_sharded or text_parse" > $OUT/pytest_new.log 2>&1
echo "new tests exit $?"; tail -60 $OUT/pytest_new.log
echo "== rest of the gpu suite"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 --deselect tests/test_gpu_parity.py::test_config1_full_size_matches_threaded_port > $OUT/pytest_gpu.log 2>&1
echo "suite exit $?"; tail -30 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
echo "== default bench"
timeout 1500 python bench.py --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench exit $?"; cat $OUT/bench_full.json; tail -8 $OUT/bench_full.err
