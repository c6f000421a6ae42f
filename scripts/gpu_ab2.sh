#!/bin/bash
# Alternating A/B of bench.py between TWO BUILDS on one box: the working tree against a copy of another revision built under _ab/<name>
# (developer tool: `git archive <rev> | tar -x -C _ab/<name>` + build there first).  TAG=.. OTHER=head ROUNDS=3 bash scripts/gpu_ab2.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=$ROOT/gpurun_out/${TAG:-ab2}; mkdir -p $O
ARGS=${BENCH_ARGS:---steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-db --no-check}
for r in $(seq 1 ${ROUNDS:-3}); do
  for vv in ${OTHER:-head} new ${NEW_VARIANTS:-}; do     # NEW_VARIANTS="name:ENV=1,ENV2=2 ..." run the working tree under that environment
    v=${vv%%:*}; envs=""; [ "$vv" != "$v" ] && envs=${vv#*:}
    if [ $v = ${OTHER:-head} ]; then d=$ROOT/_ab/$v; else d=$ROOT; fi
    for rep in a b; do     # twice in a row: consecutive processes on a box alternate between two placements of the arena (partition 20 / 23 ms)
    ( cd $d; IFS=','; for e in $envs; do [ -n "$e" ] && export "$e"; done; unset IFS; timeout 300 python bench.py $ARGS > $O/${v}_$r$rep.json 2> $O/${v}_$r$rep.err )
    python - "$O/${v}_$r$rep.json" "$v$rep" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]; sp = r.get("sort_pass", {})
    cnt = r if r.get("dominant") == "count" else r.get("kernels", {}).get("count", {})
    print("%-10s %7.2f ms  %s  pass1 %.4f pass2 %.4f count %.4f (wall %.2f)" % (sys.argv[2], d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()},
          sp.get("avg_launch_ms", 0), sp.get("second_pass", {}).get("avg_launch_ms", 0), cnt.get("avg_launch_ms", 0), cnt.get("wall_ms_per_step", 0)))
except Exception as e:
    print(sys.argv[2], "no bench line:", e)
PY
    done
  done
done
