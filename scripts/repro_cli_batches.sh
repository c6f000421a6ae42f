cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04t
python - <<'PY'
import sys; sys.path.insert(0,'.')
import oracle
bases = oracle.synth_reads(8, 200_000, 0, 40_000).tobytes()
reads = [r for r in bases.decode().split(".") if r]
open("/tmp/r.fq","w").write("".join("@%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads)))
PY
for k in 31 21; do
echo "== k=$k"
MERYL_BATCH_BASES=1500000 AMD_LOG_LEVEL=1 meryl_amd/bin/meryl -V k=$k memory=2 count /tmp/r.fq output /tmp/many$k.meryl > gpurun_out/r04t/cli_$k.out 2> gpurun_out/r04t/cli_$k.err; echo "rc $?"
grep -n -i 'hip.*error\|invalid\|ERROR' gpurun_out/r04t/cli_$k.err | head -20
done
echo "== k=31 MGC_HASH64M=0"
MGC_HASH64M=0 MERYL_BATCH_BASES=1500000 meryl_amd/bin/meryl -Q k=31 memory=2 count /tmp/r.fq output /tmp/many31b.meryl; echo "rc $?"
echo "== k=31 MGC_SOA5=0"
MGC_SOA5=0 MERYL_BATCH_BASES=1500000 meryl_amd/bin/meryl -Q k=31 memory=2 count /tmp/r.fq output /tmp/many31c.meryl; echo "rc $?"
