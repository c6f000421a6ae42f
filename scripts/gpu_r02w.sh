#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02w
mkdir -p $OUT
timeout 600 python -m pytest tests/test_db_device.py tests/test_cli.py -m gpu -x -q -k "node or gpus" 2>&1 | tail -4
timeout 600 python scripts/node_bench.py > $OUT/node_bench.json 2> $OUT/node_bench.err; cat $OUT/node_bench.json; tail -3 $OUT/node_bench.err
