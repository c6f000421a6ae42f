#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02h
mkdir -p $OUT
export TMPDIR=/tmp
hipcc -O2 scripts/ubench/pinned_write.cpp -o /tmp/pinned_write -lpthread 2>&1 | grep -v warning | tail -3
timeout 300 /tmp/pinned_write 2>&1 | tee $OUT/pinned_write.log
echo "== e2e mmap"
MGC_TEXT_MMAP=1 MGC_IO_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-check --steps 1 --warmup 1 > $OUT/bench_mmap.json 2> $OUT/bench_mmap.err
python - <<PY
import json
d = json.load(open("$OUT/bench_mmap.json"))
print(json.dumps(d.get("e2e"), indent=0))
PY
