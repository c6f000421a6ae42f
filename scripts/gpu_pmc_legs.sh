#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, kernel-trace only) of the other configs' single-GPU legs
# (scripts/kbench.py: k = 31, k = 31 compress, k = 51), per kernel and launch, scaled with the calibration of the same round's
# gpu_final.sh run (gpurun_out/${TAG}_final/pmc_traffic.json; 2.0 / 1.0 if absent).  TAG=r05 bash scripts/gpu_pmc_legs.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-r05}
O=gpurun_out/${TAG}_final; mkdir -p $O
export TMPDIR=/tmp
for a in "k31 31" "k31c 31 250000 1 20000" "k51 51"; do set -- $a; n=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/legpmc_${n}_$c -o b -- python scripts/kbench.py "$@" > /dev/null 2> $O/legpmc_${n}_$c.err
    echo "leg $n $c exit $?"
  done
done
OUT=$O python - <<'PY'
import csv, glob, collections, json, os
OUT = os.environ["OUT"]
fs, ws = 2.0, 1.0
try:
    cal = json.load(open(OUT + "/pmc_traffic.json"))["calibration"]
    fs, ws = cal["fetch_scale_read8"], cal["write_scale_copy8"]
except Exception:
    pass
res = {}
for leg in ("k31", "k31c", "k51"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob('%s/legpmc_%s_%s/**/*counter_collection.csv' % (OUT, leg, c), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r.get('Kernel_Name', '').split('(')[0].replace('void ', '').replace('mgc::', '')
                agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    ks = {}
    for k, d in sorted(agg.items()):
        if not any(x in k for x in ("radix", "hash", "kmer", "compact", "hpc")):
            continue
        f, w = d.get("FETCH_SIZE"), d.get("WRITE_SIZE")
        ks[k] = {"launches": len(f or w or []), "fetch_bytes_per_launch": (sum(f) / len(f)) * 1024 * fs if f else None,
                 "write_bytes_per_launch": (sum(w) / len(w)) * 1024 * ws if w else None}
    res[leg] = ks
json.dump({"source": "scripts/gpu_pmc_legs.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python scripts/kbench.py <leg> (1 warm-up + 3 counts of 5 Gbp); per-launch means",
           "fetch_scale": fs, "write_scale": ws, "legs": res}, open(OUT + "/pmc_traffic_legs.json", "w"), indent=1)
for leg, ks in res.items():
    for k, v in ks.items():
        if v["launches"] >= 3 and (v["fetch_bytes_per_launch"] or 0) + (v["write_bytes_per_launch"] or 0) > 5e7:
            print(leg, k[:64].ljust(64), v["launches"], "fetch %.3e" % (v["fetch_bytes_per_launch"] or 0), "write %.3e" % (v["write_bytes_per_launch"] or 0))
PY
rm -rf $O/legpmc_*_FETCH_SIZE $O/legpmc_*_WRITE_SIZE
