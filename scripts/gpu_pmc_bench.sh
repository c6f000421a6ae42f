#!/bin/bash
# HBM traffic of the bench's kernels from the TCC counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 passes, kernel-trace only, each calibrated on known-byte kernels of the same
# access width.  Writes gpurun_out/pmc_bench/summary.json.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_bench
mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib scripts/ubench/calib.hip 2> $OUT/calib_build.log || { echo "calib build failed"; cat $OUT/calib_build.log; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -o c -- /tmp/calib > $OUT/calib_$c.log 2>&1
  echo "calib $c exit $?"
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/bench_$c -o b -- python bench.py --reads ${READS:-66666667} --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  echo "bench $c exit $?"
done
python - <<'PY'
import csv, glob, collections, json
def load(tag):
    agg = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/pmc_bench/%s/**/*counter_collection.csv' % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get('Kernel_Name', '').split('(')[0].replace('void ', '').replace('mgc::', '')
            agg[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
    return agg
N8 = float(8 << 28)
out = {"calibration": {}, "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    a = load("calib_" + c)
    for (k, cn), v in sorted(a.items()):
        out["calibration"]["%s:%s" % (k, cn)] = {"launches": len(v), "mean_counter": sum(v) / len(v)}
# counter units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
def mean(a, k, c):
    v = a.get((k, c)); return sum(v) / len(v) if v else None
fr = mean(load("calib_FETCH_SIZE"), "calib_read8", "FETCH_SIZE")
fc = mean(load("calib_FETCH_SIZE"), "calib_copy8", "FETCH_SIZE")
wc = mean(load("calib_WRITE_SIZE"), "calib_copy8", "WRITE_SIZE")
out["calibration"]["bytes_per_buffer"] = N8
out["calibration"]["fetch_scale_read8"] = N8 / (fr * 1024) if fr else None      # true bytes per reported KiB*1024
out["calibration"]["fetch_scale_copy8"] = N8 / (fc * 1024) if fc else None
out["calibration"]["write_scale_copy8"] = N8 / (wc * 1024) if wc else None
fs = out["calibration"]["fetch_scale_read8"] or 1.0
ws = out["calibration"]["write_scale_copy8"] or 1.0
bf, bw = load("bench_FETCH_SIZE"), load("bench_WRITE_SIZE")
names = sorted({k for (k, _) in list(bf) + list(bw)})
for k in names:
    f, w = bf.get((k, "FETCH_SIZE")), bw.get((k, "WRITE_SIZE"))
    out["kernels"][k] = {
        "launches": len(f or w or []),
        "fetch_bytes_per_launch": (sum(f) / len(f)) * 1024 * fs if f else None,
        "write_bytes_per_launch": (sum(w) / len(w)) * 1024 * ws if w else None,
    }
json.dump(out, open('gpurun_out/pmc_bench/summary.json', 'w'), indent=1)
print(json.dumps(out["calibration"], indent=1))
for k, v in out["kernels"].items():
    if v["launches"] and ("radix" in k or "hash" in k or "kmer" in k):
        print(k[:60].ljust(60), v["launches"], "fetch %.3e" % (v["fetch_bytes_per_launch"] or 0), "write %.3e" % (v["write_bytes_per_launch"] or 0))
PY
