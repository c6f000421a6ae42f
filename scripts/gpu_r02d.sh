#!/bin/bash
# round 2, call D: host I/O rates; tile-shape / sub-bucket-size sweep of the count step; traced file -> database run
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02d
mkdir -p $OUT
export TMPDIR=/tmp
echo "== host io"; timeout 600 python scripts/hostio_bench.py 2>&1 | tee $OUT/hostio.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-check --steps 4 --warmup 1 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("$name", round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, round(d["roofline"]["frac"], 3))
except Exception as e:
    print("$name FAILED", e, open("$OUT/bench_$name.err").read()[-300:])
PY
}
run base X=0
run kpt8 MGC_SORT_KPT=8
run b512 MGC_SORT_BLOCK=512
run b512k8 MGC_SORT_BLOCK=512 MGC_SORT_KPT=8
run tgt768 MGC_FINISH_TARGET=768
run tgt1400 MGC_FINISH_TARGET=1400
run grid2048 MGC_HASH_GRID=2048
echo "== traced e2e"
MGC_IO_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-check --steps 2 --warmup 1 > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err
python - <<PY
import json
d = json.load(open("$OUT/bench_e2e.json"))
print(json.dumps(d.get("db_write"), indent=0)); print(json.dumps(d.get("e2e"), indent=0))
PY
grep "\[io\]" $OUT/bench_e2e.err | head
