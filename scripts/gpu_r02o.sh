#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02o
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --maxfail=10 --deselect tests/test_gpu_parity.py::test_config1_full_size_matches_threaded_port > $OUT/pytest.log 2>&1
echo "exit $?"; tail -6 $OUT/pytest.log
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-check --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, round(d["roofline"]["frac"], 3))
PY
MGC_HASH_DBG=1 timeout 300 python scripts/kbench.py 21 13333334 2>&1 | grep -i "hash\|phase\|cycles" | head -5
