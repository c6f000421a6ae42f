#!/bin/bash
# round 2, last call: bench line (without the CPU leg: the GPU budget of the round is nearly spent) + rocprofv3 kernel stats of the very final state
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02zz
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cut -c1-400 $OUT/bench.json
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o full -- python bench.py --no-cpu-baseline --no-e2e --no-check > $OUT/prof_bench.json 2> $OUT/prof_bench.err
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
head -5 $OUT/kernel_stats.csv | cut -c1-200
