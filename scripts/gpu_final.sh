#!/bin/bash
# Final state of a round: default bench line, rocprofv3 kernel stats of the same workload, calibrated PMC traffic.
# TAG=r02 bash scripts/gpu_final.sh  -> gpurun_out/$TAG_final/{bench_full.json,kernel_stats.csv,pmc_traffic.json}
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-r03}
OUT=gpurun_out/${TAG}_final
mkdir -p $OUT
export TMPDIR=/tmp
echo "== default bench (the driver's command)"
timeout 1500 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench exit $?"; cut -c1-2500 $OUT/bench_full.json; tail -3 $OUT/bench_full.err
echo "== rocprofv3 kernel stats of the same workload"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o full -- python bench.py --no-cpu-baseline --no-e2e --no-check --no-db > $OUT/prof_bench.json 2> $OUT/prof_bench.err
echo "rocprof exit $?"
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null; head -16 $OUT/kernel_stats.csv
echo "== PMC traffic (separate passes, kernel-trace only, calibrated)"
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib scripts/ubench/calib.hip 2> $OUT/calib_build.log || cat $OUT/calib_build.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -o c -- /tmp/calib > $OUT/calib_$c.log 2>&1
  echo "calib $c exit $?"
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/bench_$c -o b -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-check --no-db > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  echo "bench $c exit $?"
done
OUT=$OUT python - <<'PY'
import csv, glob, collections, json, os
OUT = os.environ["OUT"]
def load(tag):
    agg = collections.defaultdict(list)
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (OUT, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get('Kernel_Name', '').split('(')[0].replace('void ', '').replace('mgc::', '')
            agg[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
    return agg
N8 = float(8 << 28)
def mean(a, k, c):
    v = a.get((k, c)); return sum(v) / len(v) if v else None
cf, cw = load("calib_FETCH_SIZE"), load("calib_WRITE_SIZE")
fr, wc = mean(cf, "calib_read8", "FETCH_SIZE"), mean(cw, "calib_copy8", "WRITE_SIZE")
fs = N8 / (fr * 1024) if fr else 1.0          # true bytes per reported KiB (MI355X_MICROARCH.md: calibrate per access width)
ws = N8 / (wc * 1024) if wc else 1.0
# round 5: the same 2 GiB read with 4-, 16- and 1-byte loads, written with 4- and 1-byte stores -- the scales per width
widths = {}
for nm, ctr, agg in (("read4", "FETCH_SIZE", cf), ("read16", "FETCH_SIZE", cf), ("read1", "FETCH_SIZE", cf), ("copy4", "WRITE_SIZE", cw), ("write1", "WRITE_SIZE", cw)):
    v = mean(agg, "calib_" + nm, ctr)
    widths[nm] = {"counter_KiB": v, "scale": (N8 / (v * 1024)) if v else None}
bf, bw = load("bench_FETCH_SIZE"), load("bench_WRITE_SIZE")
kernels = {}
for k in sorted({k for (k, _) in list(bf) + list(bw)}):
    f, w = bf.get((k, "FETCH_SIZE")), bw.get((k, "WRITE_SIZE"))
    kernels[k] = {"launches": len(f or w or []),
                  "fetch_bytes_per_launch": (sum(f) / len(f)) * 1024 * fs if f else None,
                  "write_bytes_per_launch": (sum(w) / len(w)) * 1024 * ws if w else None}
dom = [k for k in kernels if k.startswith("radix_group_kernel<unsigned long long")]
out = {"source": "scripts/gpu_final.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes; python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-check)",
       "reads_per_gpu": 66666667, "kernel": dom[0] if dom else None,
       "calibration": {"bytes_per_buffer": N8, "fetch_scale_read8": fs, "write_scale_copy8": ws,
                       "calib_read8_FETCH_SIZE_KiB": fr, "calib_copy8_WRITE_SIZE_KiB": wc, "by_access_width": widths},
       "fetch_bytes_per_launch": kernels[dom[0]]["fetch_bytes_per_launch"] if dom else None,
       "write_bytes_per_launch": kernels[dom[0]]["write_bytes_per_launch"] if dom else None,
       "launches": kernels[dom[0]]["launches"] if dom else 0, "all_kernels": kernels}
json.dump(out, open(OUT + "/pmc_traffic.json", "w"), indent=1)
print("fetch scale %.3f write scale %.3f" % (fs, ws), "by width:", {k: (round(v["scale"], 3) if v["scale"] else None) for k, v in widths.items()})
for k, v in kernels.items():
    if v["launches"] and any(x in k for x in ("radix", "hash", "kmer", "compact", "encode", "merge")):
        print(k[:70].ljust(70), v["launches"], "fetch %.3e" % (v["fetch_bytes_per_launch"] or 0), "write %.3e" % (v["write_bytes_per_launch"] or 0))
PY
rm -rf $OUT/prof $OUT/calib_FETCH_SIZE $OUT/calib_WRITE_SIZE $OUT/bench_FETCH_SIZE $OUT/bench_WRITE_SIZE
ls -la $OUT
