#!/bin/bash
# PMC passes over the sort micro-bench (counters in their own runs, kernel-trace only).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc
mkdir -p $OUT
export TMPDIR=/tmp
export SORT_VARIANTS="${SORT_VARIANTS:-0:8:16:512 1:8:16:512}"
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9a-z\[\]]+|TCP_[A-Z_0-9a-z]+|FETCH_SIZE|WRITE_SIZE|GRBM_[A-Z_]+|MemUnitStalled|LdsBankConflict|VALUBusy|SALUBusy|MemUnitBusy|L2CacheHit)\b" | sort -u > $OUT/counters_available.txt
wc -l $OUT/counters_available.txt
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python scripts/sort_bench.py ${SORT_N:-135000000} 36 > $OUT/p$i.log 2>&1
  echo "pmc set $i ($set) exit $?"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '')
        if 'radix' not in name: continue
        short = name.split('(')[0].replace('void mgc::', '')
        agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc/summary.txt', 'w') as out:
    for k, d in sorted(agg.items()):
        out.write(k + '\n')
        for c, v in sorted(d.items()):
            out.write('   %-28s n=%4d  mean=%.4g  sum=%.4g\n' % (c, len(v), sum(v) / len(v), sum(v)))
print(open('gpurun_out/pmc/summary.txt').read())
PY
