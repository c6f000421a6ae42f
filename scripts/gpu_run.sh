#!/bin/bash
# One gpurun call, parameterised (replaces the per-experiment scripts of round 2):
#   TAG=r03a TESTS="tests/test_gpu_parity.py -k bitmap" BENCH="default:;nobitmap:MGC_FINISH_BITMAP=0" bash scripts/gpu_run.sh
#   TESTS   pytest arguments (empty: no tests), TESTS_K its -k expression;  BENCH  ';'-separated  name:ENV=VAL,ENV=VAL  bench.py variants
#   BENCH_ARGS  bench.py flags of the variants (default: --steps 5 --warmup 1 --no-cpu-baseline --no-e2e)
#   PROF=1  rocprofv3 --kernel-trace --stats of the default bench;  EXTRA  a shell command run at the end
# Everything is wrapped in `timeout` so a wedged kernel cannot hold the box.  Output: gpurun_out/$TAG/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -4; nproc; free -g | head -2) > $OUT/env.log 2>&1
if [ -n "${TESTS:-}" ]; then
  echo "== pytest $TESTS"
  timeout ${TEST_TIMEOUT:-900} python -m pytest $TESTS ${TESTS_K:+-k "$TESTS_K"} -m gpu ${TEST_X--x} -q -p no:cacheprovider > $OUT/pytest.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/pytest.log
  tail -${TEST_TAIL:-15} $OUT/pytest.log
fi
BENCH_ARGS=${BENCH_ARGS:---steps 5 --warmup 1 --no-cpu-baseline --no-e2e}
IFS=';' read -ra VARS <<< "${BENCH:-}"
for v in "${VARS[@]}"; do
  [ -z "$v" ] && continue
  name=${v%%:*}; envs=${v#*:}
  echo "== bench $name [$envs] $BENCH_ARGS"
  ( IFS=','; for e in $envs; do [ -n "$e" ] && export "$e"; done; unset IFS; timeout 600 python bench.py $BENCH_ARGS > $OUT/bench_$name.json 2> $OUT/bench_$name.err )
  echo "exit $?"
  python - "$OUT/bench_$name.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline", {})
    print("ms/step %.2f  value %.3e  stages %s" % (d["ms_per_step"], d["value"], {k: round(v, 2) for k, v in d.get("stage_ms_per_step", {}).items()}))
    print("  pass1 %.3f ms frac %.3f  pass2 %s  check %s" % (r.get("avg_launch_ms", 0), r.get("frac", 0),
          {k: round(v, 4) for k, v in r.get("second_pass", {}).items() if k in ("avg_launch_ms", "frac")}, d.get("check", {}).get("ok")))
except Exception as e:
    print("no bench line:", e)
PY
  tail -2 $OUT/bench_$name.err
done
if [ "${PROF:-0}" = "1" ]; then
  echo "== rocprofv3 kernel stats"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o full -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-check > $OUT/prof_bench.json 2> $OUT/prof_bench.err
  echo "rocprof exit $?"
  cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null; head -14 $OUT/kernel_stats.csv | cut -c1-200
  rm -rf $OUT/prof
fi
if [ -n "${EXTRA:-}" ]; then echo "== extra: $EXTRA"; bash -c "$EXTRA"; fi
