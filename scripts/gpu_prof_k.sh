#!/bin/bash
# kernel stats of scripts/kbench.py for another k / compress / read length (developer tool)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/profk
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o pk -- python scripts/kbench.py "$@" > $OUT/run.log 2>&1
grep "k=" $OUT/run.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/profk/pk_kernel_stats.csv')))
for r in rows[:12]:
    print(r['Name'].replace('void mgc::','').replace('mgc::','')[:64].ljust(64), r['Calls'].rjust(6), ('%.1f us avg'%(float(r['AverageNs'])/1e3)).rjust(14), ('%.1f ms tot'%(float(r['TotalDurationNs'])/1e6)).rjust(14))
PY
