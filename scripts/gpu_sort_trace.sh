#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/sorttrace
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o st -- python scripts/sort_bench.py ${SORT_N:-135000000} 36 > $OUT/run.log 2>&1
grep variant $OUT/run.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/sorttrace/st_kernel_stats.csv')))
for r in rows:
    print(r['Name'].replace('void mgc::','')[:70].ljust(70), r['Calls'].rjust(5), ('%.1f us avg'%(float(r['AverageNs'])/1e3)).rjust(14), ('min %.1f'%(float(r['MinNs'])/1e3)).rjust(12))
PY
