#!/bin/bash
# The single-GPU legs of BASELINE configs 3-5 (scripts/kbench.py: k=21 reference, k=31, k=31 compress, k=51 on 5 Gbp) under
# rocprofv3 --kernel-trace --stats -> gpurun_out/$TAG_legs/{kbench.jsonl,kernel_stats_*.csv}
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-r03}
O=gpurun_out/${TAG}_legs
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- python scripts/kbench.py "$@" > $O/kb_$name.json 2> $O/kb_$name.err; echo "$name exit $?"
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$name.csv; rm -rf $O/prof_$name
  python - $O/kb_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("  ms/step %.2f  stages %s  pass %.3f ms frac %.3f" % (d["ms_per_step"], {k: round(v,2) for k,v in d["stage_ms_per_step"].items()}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
except Exception as e: print("  no line", e)
PY
}
run k21 21
run k31 31
run k31c 31 250000 1 20000
run k51 51
run k51l 51 33333334 0 150 0 8
cat $O/kb_k21.json $O/kb_k31.json $O/kb_k31c.json $O/kb_k51.json $O/kb_k51l.json > $O/kbench.jsonl
