#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02j
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
print(build.build_cli())
PY
M=meryl_amd/bin/meryl
for hp in 1 0; do
  echo "== host parser=$hp, batches of 2 Gbases"
  MERYL_HOST_PARSER=$hp MERYL_BATCH_BASES=2000000000 MGC_IO_TRACE=1 $M -V k=51 memory=64 threads=32 count /dev/shm/ooc/r.fa output /dev/shm/ooc/out$hp.meryl 2>&1 | grep -E "\[io\]|TIMING|batches" | tee -a $OUT/ooc_cli.log
done
echo "== single pass"
MGC_IO_TRACE=1 $M -V k=51 memory=64 threads=32 count /dev/shm/ooc/r.fa output /dev/shm/ooc/one.meryl 2>&1 | grep -E "\[io\]|TIMING|batches" | tee -a $OUT/ooc_cli.log
cmp /dev/shm/ooc/one.meryl/0x000000.merylData /dev/shm/ooc/out0.meryl/0x000000.merylData && cmp /dev/shm/ooc/one.meryl/merylIndex /dev/shm/ooc/out1.meryl/merylIndex && echo "databases identical" | tee -a $OUT/ooc_cli.log
rm -rf /dev/shm/ooc
