O=gpurun_out/r03z
run() { name=$1; shift
  timeout 400 python scripts/kbench.py "$@" > $O/kb_$name.json 2> $O/kb_$name.err; echo "$name exit $?"
  python - $O/kb_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("  ms/step %.2f  stages %s  pass %.3f ms frac %.3f" % (d["ms_per_step"], {k: round(v,2) for k,v in d["stage_ms_per_step"].items()}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
except Exception as e: print("  no line", e)
PY
}
run k31c 31 250000 1 20000
MGC_HASH64_SMALL=0 run k31c_big 31 250000 1 20000
run k31 31
