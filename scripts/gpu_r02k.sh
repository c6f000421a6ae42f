#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02k
mkdir -p $OUT
export TMPDIR=/tmp
python - <<'PY'
import sys, os
sys.path.insert(0, '.')
import torch
from meryl_amd import count, build
reads, L = 40_000_000, 150
bases = count.dev_synth_reads(5, reads * L // 50, 0, reads, L, 5000, 100)
os.makedirs("/dev/shm/ooc", exist_ok=True)
with open("/dev/shm/ooc/r.fa", "wb") as f:
    step = 4_000_000
    for a in range(0, reads, step):
        n = min(step, reads - a)
        r = torch.empty((n, L + 3), dtype=torch.uint8, device="cuda")
        r[:, 0] = ord(">"); r[:, 1] = 10
        r[:, 2:2 + L] = bases[a * (L + 1):(a + n) * (L + 1)].view(n, L + 1)[:, :L]
        r[:, 2 + L] = 10
        f.write(r.cpu().numpy().tobytes())
PY
M=meryl_amd/bin/meryl
for k in 51 21 51; do
  echo "== k=$k single pass"
  MGC_IO_TRACE=1 MGC_FINISH_TRACE=1 $M -V k=$k memory=64 threads=32 count /dev/shm/ooc/r.fa output /dev/shm/ooc/one.meryl 2>&1 | grep -E "host pushes|TIMING|finish\]" | head -8 | tee -a $OUT/cli.log
  rm -rf /dev/shm/ooc/one.meryl
done
rm -rf /dev/shm/ooc
