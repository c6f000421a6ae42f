O=gpurun_out/r03v
for v in "51 1" "51 0" "31 1" "31 0" "35 1"; do set -- $v
  MGC_HASH_BINRANK=$2 timeout 300 python scripts/kbench.py $1 > $O/kb_$1_b$2.json 2> $O/kb_$1_b$2.err; echo "k=$1 binrank=$2 exit $?"
  python - $O/kb_$1_b$2.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("  ms/step %.2f  stages %s  pass %.3f ms frac %.3f" % (d["ms_per_step"], {k: round(v,2) for k,v in d["stage_ms_per_step"].items()}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
except Exception as e: print("  no line", e)
PY
done
