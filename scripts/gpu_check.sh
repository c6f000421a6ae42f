#!/bin/bash
# One gpurun call: parity tests, smoke, benches, rocprof summary.
# Everything is wrapped in `timeout` so a wedged kernel cannot hold the box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2) > $OUT/env.log 2>&1

if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
fi

READS=${READS:-13333334}    # 2 Gbp
VARIANTS=${VARIANTS:-"0:8:16 0:9:16"}
for cfg in $VARIANTS; do
  IFS=: read m rb kpt <<< "$cfg"
  echo "== bench reads=$READS mode=$m rb=$rb kpt=$kpt"
  MGC_SORT_MODE=$m MGC_RADIX_BITS=$rb MGC_SORT_KPT=$kpt timeout 600 python bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline \
     > $OUT/bench_m${m}_rb${rb}_kpt${kpt}.json 2> $OUT/bench_m${m}_rb${rb}_kpt${kpt}.err
  echo "exit $?"; cut -c1-1800 $OUT/bench_m${m}_rb${rb}_kpt${kpt}.json; tail -3 $OUT/bench_m${m}_rb${rb}_kpt${kpt}.err
done

if [ "${FULL:-1}" = "1" ]; then
echo "== full-size default bench (BASELINE configs[1]: 10 Gbp)"
timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "exit $?"; cat $OUT/bench_full.json; tail -5 $OUT/bench_full.err
echo "== rocprof kernel stats of the same command"
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o full -- python bench.py --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof_bench.err
echo "rocprof exit $?"
ls $OUT/prof | head; head -20 $OUT/prof/*kernel_stats.csv
fi
