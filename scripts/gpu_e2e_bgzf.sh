#!/bin/bash
# File -> database wall clock for compressed inputs: one gzip stream against BGZF (block-parallel inflate), FASTA and BAM.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/e2e
mkdir -p $OUT
READS=${READS:-4000000}
python - <<PY
import sys, time, struct, zlib, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle
reads = $READS
b = oracle.synth_reads(2, 20_000_000, 0, reads, 150, 5000, 100)
rows = b.reshape(reads, 151)[:, :150]
rec = np.empty((reads, 3 + 150 + 1), dtype=np.uint8)
rec[:, :3] = np.frombuffer(b'>r\n', dtype=np.uint8); rec[:, 3:153] = rows; rec[:, 153] = 10
text = rec.tobytes()
def bgzf_fast(data, block=0xff00):
    out = []
    for i in range(0, len(data), block):
        d = data[i:i + block]
        c = zlib.compressobj(1, zlib.DEFLATED, -15); cd = c.compress(d) + c.flush()
        bs = 12 + 6 + len(cd) + 8
        out.append(struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + cd +
                   struct.pack("<II", zlib.crc32(d) & 0xffffffff, len(d)))
    return b"".join(out)
t = time.time()
open('/tmp/e2e_bgzf.fasta.gz', 'wb').write(bgzf_fast(text))
# BAM: 4-bit bases, no qualities worth speaking of (0xff), one record per read
lut = np.zeros(256, dtype=np.uint8)
for i, ch in enumerate(b"=ACMGRSVTWYHKDBN"): lut[ch] = i
codes = lut[rows]
packed = (codes[:, 0::2] << 4) | codes[:, 1::2]                                    # 150 bases -> 75 bytes
fixed = struct.pack("<iiBBHHHiiii", -1, -1, 2, 0, 4680, 0, 4, 150, -1, -1, 0) + b"r\0"
recb = np.empty((reads, 4 + len(fixed) + 75 + 150), dtype=np.uint8)
recb[:, :4] = np.frombuffer(struct.pack("<i", len(fixed) + 75 + 150), dtype=np.uint8)
recb[:, 4:4 + len(fixed)] = np.frombuffer(fixed, dtype=np.uint8)
recb[:, 4 + len(fixed):4 + len(fixed) + 75] = packed
recb[:, 4 + len(fixed) + 75:] = 0xff
hdr = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 0)
open('/tmp/e2e.bam', 'wb').write(bgzf_fast(hdr + recb.tobytes()))
open('/tmp/e2e.fasta', 'wb').write(text)
print('wrote inputs in %.1f s' % (time.time() - t))
PY
gzip -1 -k -f /tmp/e2e.fasta
python -m meryl_amd.build > /dev/null 2>&1
for f in /tmp/e2e.fasta /tmp/e2e.fasta.gz /tmp/e2e_bgzf.fasta.gz /tmp/e2e.bam; do
  rm -rf /tmp/e2e.meryl
  t0=$(date +%s.%N)
  meryl_amd/bin/meryl -V k=21 memory=32 threads=16 count $f output /tmp/e2e.meryl 2> $OUT/cli_fmt.log
  echo "$(basename $f) ($(du -m $f | cut -f1) MB): exit $? wall $(python3 -c "import sys,time; print('%.3f' % (time.time() - float(sys.argv[1])))" $t0) s  $(grep -E "TIMING" $OUT/cli_fmt.log)  $(grep -E "distinct k-mers" $OUT/cli_fmt.log | tail -1)"
done
