#!/usr/bin/env python3
"""Census of FLAT memory operations (flat_load / flat_store) in the gfx950 ISA hipcc emits for csrc/*.hip.  A kernel that only touches global memory
and LDS should have none: a flat operation means the compiler lost the address space — e.g. `cond ? p[i] : fill` with a run-time `fill`, which it
compiles to a select between `&p[i]` and a stack slot holding `fill` (r06: delta16_bwd_kernel's staging, lstm16_train_kernel's checkpoint loads:
a scratch store + flat loads per use, 16 - 32 B of scratch per lane).  Expected survivors: the sweep kernels (their pointers come from a run table).
usage: tools/flat_ops_census.py [file.hip ...]      (default: every csrc/*.hip, with the flags build.py uses)"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def own_flags(src):
    out = []
    with open(src) as f:
        for _, ln in zip(range(60), f):
            if ln.startswith("// odpd-build-flags:"):
                out += ln.split(":", 1)[1].split()
    return out


def scan(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only", "-S", src, "-o", t.name,
                               *own_flags(src)], stderr=subprocess.DEVNULL)
        text = open(t.name).read()
    kernel, found = "?", {}
    for ln in text.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
        if "flat_load" in ln or "flat_store" in ln or "flat_atomic" in ln:
            found[kernel] = found.get(kernel, 0) + 1
    return os.path.basename(src), found


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "opendpd_amd", "csrc", "*.hip")))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        for name, found in pool.map(scan, files):
            other = {k: v for k, v in found.items() if "sweep" not in k}
            print(f"{name}: {len(found)} kernel(s) with flat operations, {len(other)} of them not sweep kernels")
            for k, v in sorted(other.items()):
                print(f"   {v:3d}  {k[:110]}")


if __name__ == "__main__":
    main()
