#!/bin/bash
# SQ / GRBM counters of every kernel of the judged step (rocprofv3 --pmc, counters in their own passes with --kernel-trace only):
# where a kernel's wave cycles go (issuing VALU / LDS, parked on s_waitcnt or a barrier, stalled at issue), instructions per
# class, LDS bank conflicts.  TAG=r03k bash scripts/gpu_pmc_kernels.sh  ->  gpurun_out/$TAG/pmc_sq.json (+ .txt)
# CMD overrides the profiled command (default: one step of the 10 Gbp bench).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD=${CMD:-python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-check --no-db}
i=0
for set in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
 "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" \
 "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1
  echo "pmc set $i ($set) exit $?"
done
OUT=$OUT CMD_DESC="$CMD" python - <<'PY'
import csv, glob, collections, json, os
OUT = os.environ["OUT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(OUT + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '').split('(')[0].replace('void ', '').replace('mgc::', '')
        if not any(x in name for x in ("radix", "hash", "bitmap", "kmer", "compact", "narrow", "subbucket")):
            continue
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    m["launches"] = max(len(v) for v in d.values())
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:                      # quad-cycles; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md)
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in m:
                m["frac_" + c] = m[c] / wc
    if m.get("SQ_INSTS_VALU") and m.get("SQ_WAVES"):
        m["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
    if m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_bank_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    res[k] = m
json.dump({"source": "scripts/gpu_pmc_kernels.sh: rocprofv3 --kernel-trace --pmc <SQ sets>, each set in its own run of `%s`; per-launch means" % os.environ.get("CMD_DESC", "python bench.py --steps 1 --warmup 0 (the 10 Gbp step)"),
           "kernels": res}, open(OUT + "/pmc_sq.json", "w"), indent=1)
with open(OUT + "/pmc_sq.txt", "w") as out:
    for k, m in res.items():
        out.write(k[:110] + "\n")
        for c in sorted(m):
            out.write("   %-28s %.5g\n" % (c, m[c]))
print(open(OUT + "/pmc_sq.txt").read()[:6000])
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
