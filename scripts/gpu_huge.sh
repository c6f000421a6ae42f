#!/bin/bash
# per-case time of the streaming hash-count kernels (developer tool)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for args in "$@"; do
  rm -rf gpurun_out/hb; mkdir -p gpurun_out/hb
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hb -o hb -- python scripts/huge_bench.py $args 2>&1 | grep "k="
  python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/hb/hb_kernel_stats.csv')):
    if 'huge' in r['Name'] or 'hash_count' in r['Name']:
        print('   ', r['Name'][:60].ljust(60), r['Calls'], '%.1f us avg' % (float(r['AverageNs'])/1e3), '%.1f us max' % (float(r['MaxNs'])/1e3))
PY
done
