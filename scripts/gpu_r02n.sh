#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02n
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity (64-bit suffix kernels: k=28..32, compress)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --maxfail=10 -k "random_inputs or golden or oversized or capacity or compress or 31 or 32 or 28" > $OUT/pytest.log 2>&1
echo "exit $?"; tail -6 $OUT/pytest.log
for hi in 1 0; do
for args in "31 33333334" "31 250000 1 20000" "51 33333334"; do
  MGC_FINISH_HASH64I=$hi timeout 600 python scripts/kbench.py $args 2>/dev/null | tee -a $OUT/kbench_i$hi.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('idx64=$hi', d['config']['workload'][:60], '| ms/step %.1f ms/Gbp %.1f' % (d['ms_per_step'], d['ms_per_Gbp']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()})"
done
done
