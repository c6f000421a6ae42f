#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02e
mkdir -p $OUT
export TMPDIR=/tmp
hipcc -O2 scripts/ubench/ring_read.cpp -o /tmp/ring_read -lpthread 2>&1 | tail -3
head -c 12000000000 /dev/urandom > /dev/shm/ringtest.bin
for cfg in "1 0 8" "6 0 8" "6 1 8" "16 1 24" "16 0 24" "32 1 40"; do /tmp/ring_read /dev/shm/ringtest.bin $cfg; done 2>&1 | tee $OUT/ring.log
rm -f /dev/shm/ringtest.bin
echo "== db device tests (interleaved pieces)"
timeout 900 python -m pytest tests/test_db_device.py tests/test_cli.py -m gpu -q -p no:cacheprovider --maxfail=10 > $OUT/pytest_db.log 2>&1; echo "exit $?"; tail -5 $OUT/pytest_db.log
echo "== traced e2e"
MGC_IO_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-check --steps 2 --warmup 1 > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err
python - <<PY
import json
d = json.load(open("$OUT/bench_e2e.json"))
print(json.dumps(d.get("db_write"), indent=0)); print(json.dumps(d.get("e2e"), indent=0))
PY
grep "\[io\]" $OUT/bench_e2e.err | head
