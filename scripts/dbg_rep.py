import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import oracle
from meryl_amd import capi, count
args = dict(read_len=150, sub_rate_ppm=5000, n_rate_ppm=100, repeat_ppm=100_000, repeat_unit=300, repeat_families=50)
want = oracle.synth_reads(5, 3_000_000, 7, 60_000, **args)
got = count.dev_synth_reads(5, 3_000_000, 7, 60_000, **args)
print("gen equal", np.array_equal(got.cpu().numpy(), want))
k = 21
_, wlo, wcn, wni = oracle.count_brute(want.tobytes(), k)
for stream in ("1", "0"):
    os.environ["MGC_FINISH_STREAM"] = stream
    cfg = capi.configure(k, want.size, 1 << 30)
    with count.Session(cfg) as s:
        s.push_bases_device(got)
        s.count()
        klo, counts, _ = s.result()
        info = s.info()
    print("stream", stream, "n_inst", info.n_instances, wni, "nd", len(klo), len(wlo), "keys eq", np.array_equal(klo, wlo), "counts eq", np.array_equal(counts, wcn) if len(klo)==len(wlo) else None, "sum", int(counts.astype(np.int64).sum()), int(wcn.astype(np.int64).sum()), "max", counts.max(), wcn.max())
    if len(klo) == len(wlo) and not np.array_equal(counts, wcn):
        bad = np.nonzero(counts != wcn)[0]
        print("  mismatches", len(bad), "first", bad[:5], counts[bad[:5]], wcn[bad[:5]], [hex(int(x)) for x in klo[bad[:5]]])
    elif len(klo) != len(wlo):
        sw = set(wlo.tolist()); sg = set(klo.tolist())
        print("  missing", len(sw - sg), "extra", len(sg - sw), "dups in got", len(klo) - len(sg))
