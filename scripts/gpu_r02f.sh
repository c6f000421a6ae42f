#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/e2e_cli.py 2>&1 | grep -v amdgpu.ids | tee $OUT/e2e_cli.log
