#!/bin/bash
# per-kernel stats of the k = 51 leg (scripts/kbench.py 51) with K96 records and with whole 16-byte keys -> gpurun_out/$TAG/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${TAG:-r05j}; mkdir -p $O
for v in new old; do
  if [ $v = old ]; then export MGC_K96=0; else unset MGC_K96; fi
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -o p -- python scripts/kbench.py 51 > $O/kb_$v.json 2> $O/kb_$v.err; echo "exit $?"
  f=$(find $O/p_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; head -12 $f | cut -d, -f1-4 | sed 's/(.*"/"/' | cut -c1-160
  cp $f $O/kernel_stats_k51_$v.csv; rm -rf $O/p_$v
done
