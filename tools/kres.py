#!/usr/bin/env python
"""Per-kernel register / LDS / scratch figures of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line each."""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result"]
src = sys.argv[1]
p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                   capture_output=True, text=True)
cur = None
for line in p.stderr.splitlines():
    m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name)[:70]}
        continue
    m = re.search(r"(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
        if m.group(1).startswith("LDS"):
            print("%-72s sgpr %3d vgpr %3d agpr %3d scratch %4d occ %d spill v%d lds %d" % (
                cur["name"], cur.get("TotalSGPRs", -1), cur.get("VGPRs", -1), cur.get("AGPRs", -1), cur.get("ScratchSize", -1),
                cur.get("Occupancy", -1), cur.get("VGPRs Spill", -1), cur.get("LDS Size", -1)))
            cur = None
if p.returncode != 0:
    sys.stderr.write(p.stderr[-3000:])
    sys.exit(p.returncode)
