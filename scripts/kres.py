#!/usr/bin/env python3
"""Register / scratch / LDS use of the kernels whose mangled name contains PATTERN (developer tool).
Usage: python scripts/kres.py PATTERN"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = ""
for name in ("mgc_kmer.hip", "mgc_sort.hip", "mgc_scan.hip", "mgc_finish.hip", "mgc_misc.hip", "mgc_parse.hip"):
    src = os.path.join(root, "meryl_amd", "csrc", name)
    out += subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Rpass-analysis=kernel-resource-usage",
                           "-o", "/tmp/kres.o", src], capture_output=True, text=True).stderr
cur = None
for line in out.splitlines():
    if "error:" in line:
        print(line)
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        continue
    if cur and sys.argv[1] in cur:
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m:
            print(cur[:70], m.group(1), m.group(2))
