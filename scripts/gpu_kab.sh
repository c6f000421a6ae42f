#!/bin/bash
# A/B of kbench.py legs under environment variants on ONE box:  TAG=r05h LEGS="k51:51;k31:31" AB="new:;old:MGC_K96=0" bash scripts/gpu_kab.sh
# a leg is  name:kbench-arguments ; prints ms/step and the stage times of every run, keeps the bench lines in gpurun_out/$TAG/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${TAG:-kab}; mkdir -p $O
IFS=';' read -ra LG <<< "${LEGS:-k21:21;k31:31;k51:51}"
IFS=';' read -ra VARS <<< "${AB:-new:}"
for r in $(seq 1 ${ROUNDS:-1}); do
 for l in "${LG[@]}"; do
  lname=${l%%:*}; largs=${l#*:}
  for v in "${VARS[@]}"; do
    name=${v%%:*}; envs=${v#*:}
    ( IFS=','; for e in $envs; do [ -n "$e" ] && export "$e"; done; unset IFS; timeout 300 python scripts/kbench.py $largs > $O/${lname}_${name}_$r.json 2> $O/${lname}_${name}_$r.err )
    python - "$O/${lname}_${name}_$r.json" "$lname $name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-16s %7.2f ms  %6.2f ms/Gbp  %s  pass %.4f ms frac %.3f" % (sys.argv[2], d["ms_per_step"], d["ms_per_Gbp"], {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()},
          d.get("roofline", {}).get("avg_launch_ms", 0), d.get("roofline", {}).get("frac", 0)))
except Exception as e:
    print(sys.argv[2], "no bench line:", e)
PY
  done
 done
done
