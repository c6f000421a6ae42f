"""File -> database runs of the stand-alone CLI on one synthetic FASTQ (the bench's e2e workload), under several I/O settings.
usage: python scripts/e2e_cli.py [reads] -- writes /dev/shm/mgc_e2e/reads.fq once, runs `meryl count` per setting."""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from meryl_amd import build, count  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 66_666_667
L = 150
d = "/dev/shm/mgc_e2e"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)
bases = count.dev_synth_reads(2, 333_333_334, 0, reads, L, 5000, 100)
rec = 2 * L + 7
fq = os.path.join(d, "reads.fq")
with open(fq, "wb") as f:
    step = 4_000_000
    for a in range(0, reads, step):
        n = min(step, reads - a)
        r = torch.empty((n, rec), dtype=torch.uint8, device=bases.device)
        r[:, 0] = ord("@"); r[:, 1] = ord("r"); r[:, 2] = 10
        r[:, 3:3 + L] = bases[a * (L + 1):(a + n) * (L + 1)].view(n, L + 1)[:, :L]
        r[:, 3 + L] = 10; r[:, 4 + L] = ord("+"); r[:, 5 + L] = 10
        r[:, 6 + L:6 + 2 * L] = ord("I"); r[:, 6 + 2 * L] = 10
        f.write(r.cpu().numpy().tobytes())
del bases
torch.cuda.empty_cache()
if os.environ.get("E2E_WARM") == "1":                        # read the file once before anybody times anything
    t0 = time.perf_counter()
    subprocess.run("cat %s > /dev/null" % fq, shell=True)
    print("warm read of the file: %.2f s" % (time.perf_counter() - t0))
if os.environ.get("E2E_SETTLE") == "1":                      # does a big allocation in THIS process absorb the next process's slow first hipMalloc?
    t0 = time.perf_counter()
    x = torch.empty(110 << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    del x
    torch.cuda.empty_cache()
    print("settle: 110 GB allocated in %.3f s, freed in %.3f s" % (t1 - t0, time.perf_counter() - t1))
cli = build.build_cli()
# every setting: (environment additions, command prefix)
settings = [({}, [])]      # (rounds 2-3 swept the database image / slot sizes and the reader ring here: profiles/r03o_*, r03p_*; the switches are gone)
if os.environ.get("E2E_SETTINGS"):                            # "A=1,B=2;C=3;" -> one run per ';'-separated environment set (empty = defaults)
    settings = [(dict(kv.split("=", 1) for kv in grp.split(",") if kv), []) for grp in os.environ["E2E_SETTINGS"].split(";")]
if os.environ.get("E2E_MORE") == "1":
    settings += [
                 ({}, ["taskset", "-c", "0-63,128-191"]), ({}, ["taskset", "-c", "64-127,192-255"])]
try:
    print(subprocess.run(["numactl", "-H"], capture_output=True, text=True).stdout[:1500])
except OSError:
    print(subprocess.run(["lscpu"], capture_output=True, text=True).stdout[-900:])
for env_add, prefix in settings:
    time.sleep(float(os.environ.get("E2E_SETTLE_S", "6")))   # the driver clears what the previous process released; let it
    out = os.path.join(d, "out.meryl")
    shutil.rmtree(out, ignore_errors=True)
    env = dict(os.environ, MGC_IO_TRACE="1", **env_add)
    t0 = time.perf_counter()
    p = subprocess.run(prefix + [cli, "-V", "k=21", "memory=64", "threads=32", "n=10000000000", "count", fq, "output", out],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    wall = time.perf_counter() - t0
    print("==", env_add, " ".join(prefix), "rc", p.returncode, "wall %.3f s" % wall)
    for l in p.stderr.splitlines():
        if l.startswith("[io] text") or l.startswith("TIMING") or "batches" in l:
            print("   ", l[:330])
shutil.rmtree(d, ignore_errors=True)
