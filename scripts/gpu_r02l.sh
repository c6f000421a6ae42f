#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02l
mkdir -p $OUT
export TMPDIR=/tmp
echo "== compress tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -m gpu -q -p no:cacheprovider --maxfail=10 -k "compress or hpc or homopoly or out_of_core" > $OUT/pytest.log 2>&1
echo "exit $?"; tail -15 $OUT/pytest.log
echo "== kbench compress"
for args in "31 250000 1 20000" "31 33333334 1" "21 250000 1 20000" "51 250000 1 20000"; do
  for hd in 1 0; do
  MGC_HPC_DIGITS=$hd timeout 600 python scripts/kbench.py $args 2>/dev/null | tee -a $OUT/kbench_hd$hd.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('hpc_digits=$hd', d['config']['workload'][:64], '| ms/step %.1f ms/Gbp %.1f' % (d['ms_per_step'], d['ms_per_Gbp']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}, 'frac %.3f' % d.get('roofline',{}).get('frac',0), d['config']['n_distinct'])"
  done
done
