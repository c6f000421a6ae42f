#!/bin/bash
# round 2, call G: lookup + text-ring tests, e2e with the new reader defaults, config 4/5 single-GPU legs, out-of-core at 6 Gbp
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02g
mkdir -p $OUT
export TMPDIR=/tmp
echo "== lookup / text ring / cli tests"
timeout 900 python -m pytest tests/test_lookup.py tests/test_cli.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --maxfail=10 -k "lookup or cli or text_file or text_parse or out_of_core" > $OUT/pytest_new.log 2>&1
echo "exit $?"; tail -30 $OUT/pytest_new.log
echo "== traced e2e"
MGC_IO_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-check --steps 2 --warmup 1 > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err
python - <<PY
import json
d = json.load(open("$OUT/bench_e2e.json"))
print(d["ms_per_step"]); print(json.dumps(d.get("db_write"))); print(json.dumps(d.get("e2e"), indent=0))
PY
echo "== k legs (5 Gbp each)"
for args in "21 33333334" "31 33333334" "31 250000 1 20000" "31 33333334 1" "51 33333334" "51 33333334 0 150 100000"; do
  timeout 600 python scripts/kbench.py $args 2>/dev/null | tee -a $OUT/kbench.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:70], '| ms/step %.1f ms/Gbp %.1f' % (d['ms_per_step'], d['ms_per_Gbp']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}, 'frac %.3f' % d.get('roofline',{}).get('frac',0))"
done
echo "== out-of-core k=51, 3 batches of 2 Gbp"
timeout 900 python scripts/ooc_bench.py 40000000 2000000000 51 2> $OUT/ooc.err | tee $OUT/ooc.json; tail -3 $OUT/ooc.err
