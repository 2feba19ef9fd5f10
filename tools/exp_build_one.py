#!/usr/bin/env python3
"""Experiment library = the in-tree objects with ONE source recompiled under extra flags (seconds instead of a full rebuild).
usage: tools/exp_build_one.py <source.hip> <out.so> [flags ...]     (run the in-tree build first)"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, out, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
base = os.path.basename(src)[:-4]
objdir = os.path.join(ROOT, "build", "obj", "default")
objs = [o for o in sorted(glob.glob(os.path.join(objdir, "*.o"))) if os.path.basename(o) != base + ".o"]
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
obj = os.path.abspath(out) + "." + base + ".o"
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-Rpass-analysis=kernel-resource-usage", *flags, "-c", src, "-o", obj]
p = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
if p.returncode:
    sys.exit(p.stderr[-3000:])
key = os.environ.get("EXP_KERNEL", "")
name = None
for ln in p.stderr.splitlines():
    if "Function Name" in ln:
        name = ln.split("Function Name:")[1].split()[0]
    if name and key in name and ("VGPRs:" in ln or "ScratchSize" in ln):
        print(name[:70], ln.split("remark:")[1].split("[-R")[0].strip())
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, obj, "-ldl", "-o", out])
print("built", out)
