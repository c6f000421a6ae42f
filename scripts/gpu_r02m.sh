#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02m
mkdir -p $OUT
export TMPDIR=/tmp
echo "== k>32 parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --maxfail=10 -k "random_inputs or golden or oversized or capacity or compress or wide or 51 or 40 or 64 or repeat" > $OUT/pytest.log 2>&1
echo "exit $?"; tail -8 $OUT/pytest.log
echo "== kbench k=51"
for grid in 1024; do
for args in "51 33333334" "40 33333334" "51 33333334 0 150 100000"; do
  MGC_HASH_GRID=$grid timeout 600 python scripts/kbench.py $args 2>/dev/null | tee -a $OUT/kbench_g$grid.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('grid=$grid', d['config']['workload'][:60], '| ms/step %.1f ms/Gbp %.1f' % (d['ms_per_step'], d['ms_per_Gbp']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
done
done
