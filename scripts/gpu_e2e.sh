#!/bin/bash
# End-to-end (file -> database on disk) through the stand-alone CLI, plus the PCIe-inclusive rate of the host
# boundary (mgc_push_bases).  Numbers go to DESIGN.md section 8; not part of the judged bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/e2e
mkdir -p $OUT
READS=${READS:-4000000}
python - <<PY
import sys, time, numpy as np
sys.path.insert(0, '.')
import oracle
reads = $READS
t = time.time()
b = oracle.synth_reads(2, 20_000_000, 0, reads, 150, 5000, 100)          # 151 bytes per read incl. the '.' breaker
rows = b.reshape(reads, 151)[:, :150]
with open('/tmp/e2e.fasta', 'wb') as f:
    hdr = np.frombuffer(b'>r\n', dtype=np.uint8)
    rec = np.empty((reads, 3 + 150 + 1), dtype=np.uint8)
    rec[:, :3] = hdr; rec[:, 3:153] = rows; rec[:, 153] = 10
    f.write(rec.tobytes())
print('wrote /tmp/e2e.fasta: %d reads, %.2f Gbp, %.1f s' % (reads, reads * 150 / 1e9, time.time() - t))
PY
python -m meryl_amd.build > /dev/null 2>&1
for T in 1 16; do
  rm -rf /tmp/e2e.meryl
  t0=$(date +%s.%N)
  meryl_amd/bin/meryl -V k=21 memory=32 threads=$T count /tmp/e2e.fasta output /tmp/e2e.meryl 2> $OUT/cli_t$T.log
  echo "cli threads=$T exit $? wall $(python3 -c "import sys,time; print('%.3f' % (time.time() - float(sys.argv[1])))" $t0) s"
  grep -E "TIMING|Configured|bases," $OUT/cli_t$T.log
done
du -sh /tmp/e2e.meryl | cut -f1
gzip -1 -k -f /tmp/e2e.fasta
rm -rf /tmp/e2e_gz.meryl
t0=$(date +%s.%N)
meryl_amd/bin/meryl -V k=21 memory=32 threads=16 count /tmp/e2e.fasta.gz output /tmp/e2e_gz.meryl 2> $OUT/cli_gz.log
echo "cli gz threads=16 exit $? wall $(python3 -c "import sys,time; print('%.3f' % (time.time() - float(sys.argv[1])))" $t0) s"
grep -E "TIMING" $OUT/cli_gz.log
python - <<PY
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from meryl_amd import capi, count
import oracle
reads = $READS
b = oracle.synth_reads(2, 20_000_000, 0, reads, 150, 5000, 100).tobytes()
cfg = capi.configure(21, len(b), 32 << 30)
for rep in range(2):
    with count.Session(cfg, 0) as s:
        t0 = time.perf_counter()
        s.push_bases(b, end_of_sequence=False)
        t1 = time.perf_counter()
        s.count()
        t2 = time.perf_counter()
        info = s.info()
    print('host boundary: push %.3f s (%.2f GB/s host copy), upload+count %.3f s -> %.3f G distinct/s PCIe-inclusive (%d distinct)'
          % (t1 - t0, len(b) / 1e9 / (t1 - t0), t2 - t1, info.n_distinct / 1e9 / (t2 - t0), info.n_distinct))
PY
