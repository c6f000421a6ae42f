#!/bin/bash
# Alternating A/B of bench.py under environment variants on ONE box:  TAG=r05d AB="a:;b:MGC_X=0" ROUNDS=3 bash scripts/gpu_ab.sh
# prints ms/step, the stage times and the per-launch times of the two passes and the count kernel for every run.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${TAG:-ab}; mkdir -p $O
ARGS=${BENCH_ARGS:---steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-db --no-check}
IFS=';' read -ra VARS <<< "${AB}"
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in "${VARS[@]}"; do
    name=${v%%:*}; envs=${v#*:}
    ( IFS=','; for e in $envs; do [ -n "$e" ] && export "$e"; done; unset IFS; timeout 300 python bench.py $ARGS > $O/${name}_$r.json 2> $O/${name}_$r.err )
    python - "$O/${name}_$r.json" "$name" <<'PY'
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
