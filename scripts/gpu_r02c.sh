#!/bin/bash
# round 2, call C: PCIe rates; histogram-ahead A/B; parity of the new default
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pcie"; timeout 300 python scripts/pcie_bench.py 2>&1 | tee $OUT/pcie.log
for ha in 0 1; do
  echo "== bench MGC_HIST_AHEAD=$ha"
  MGC_HIST_AHEAD=$ha timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-check --steps 5 --warmup 2 > $OUT/bench_ha$ha.json 2> $OUT/bench_ha$ha.err
  echo "exit $?"; python - <<PY
import json
d = json.load(open("$OUT/bench_ha$ha.json"))
print(d["ms_per_step"], d["stage_ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
echo "== gpu suite (new default)"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 --deselect tests/test_gpu_parity.py::test_config1_full_size_matches_threaded_port > $OUT/pytest_gpu.log 2>&1
echo "suite exit $?"; tail -8 $OUT/pytest_gpu.log
