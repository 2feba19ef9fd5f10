#!/usr/bin/env python3
"""Compiles a .hip file for gfx950 and prints VGPR/AGPR/scratch/occupancy per kernel.
usage: tools/resource_usage.py file.hip [filter] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", src, "-o",
                      "/tmp/_ru.o", "-Rpass-analysis=kernel-resource-usage", *extra], capture_output=True, text=True)
txt = out.stderr
if out.returncode:
    print(txt[-3000:])
    sys.exit(1)
cur = {}
rows = []
for ln in txt.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split()[0]] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("odpd::", "").replace("(SeqArgs)", "").replace("void ", "")
    if flt in name:
        print(f"{name:60s} VGPR {str(r.get('VGPRs')):>4s} AGPR {str(r.get('AGPRs')):>4s} SGPR {str(r.get('SGPRs')):>4s} scratch {str(r.get('ScratchSize')):>4s} occ {r.get('Occupancy')}")
