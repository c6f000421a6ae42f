#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02i
mkdir -p $OUT
export TMPDIR=/tmp
MGC_IO_TRACE=1 timeout 900 python scripts/ooc_bench.py 40000000 2000000000 51 2> $OUT/ooc.err | tee $OUT/ooc.json; grep "\[io\]" $OUT/ooc.err
