#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/profsmall
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ps -- python bench.py --reads ${READS:-13333334} --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cut -c1-400 $OUT/bench.json
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/profsmall/ps_kernel_stats.csv')))
for r in rows[:16]:
    print(r['Name'].replace('void mgc::','').replace('mgc::','')[:64].ljust(64), r['Calls'].rjust(6), ('%.1f us avg'%(float(r['AverageNs'])/1e3)).rjust(14), ('%.1f ms tot'%(float(r['TotalDurationNs'])/1e6)).rjust(14))
PY
