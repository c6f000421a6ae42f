#!/bin/bash
# The two micro-benchmarks of the grouping pass: scatter (run-start alignment, output footprint) and rank
# (LDS phases only).  TAG=r05a bash scripts/gpu_ubench.sh -> gpurun_out/$TAG/{scatter,rank}_ubench.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${TAG:-r05a}; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/scatter scripts/ubench/scatter.hip && timeout 120 /tmp/scatter > $O/scatter_ubench.txt 2>&1; echo "scatter exit $?"
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/rank scripts/ubench/rank.hip && timeout 120 /tmp/rank > $O/rank_ubench.txt 2>&1; echo "rank exit $?"
cat $O/scatter_ubench.txt $O/rank_ubench.txt
