#!/bin/bash
# round 2, call V: the other k / read shapes (single-GPU legs of configs 3-5) and the in-process node count on one device
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02v
mkdir -p $OUT
: > $OUT/kbench.jsonl
for args in "21 33333334 0 150 100000" "21 33333334" "31 33333334" "31 250000 1 20000" "51 33333334 0 150 0 8"; do
  timeout 600 python scripts/kbench.py $args 2>/dev/null | tee -a $OUT/kbench.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$args', '->', round(d['ms_per_step'], 1), 'ms', {k: round(v, 1) for k, v in d['stage_ms_per_step'].items()}, 'pass frac', round(d.get('roofline', {}).get('frac', 0), 3))"
done
timeout 600 python scripts/node_bench.py > $OUT/node_bench.json 2> $OUT/node_bench.err; cat $OUT/node_bench.json; tail -3 $OUT/node_bench.err
