#!/usr/bin/env python3
"""Census of the gfx950 ISA hipcc emits for csrc/*.hip: v_mfma instructions whose DESTINATION range overlaps their SrcC range without being identical
to it (and, counted only, destinations over an A / B operand).  The compiler allows both for the four-register 16x16 forms, and the kernels run with
dozens of such allocations — exact-fp32 (`lstm16_train_kernel`, `qat16*_bwd_kernel`, ...) and bf16 (`gru16x_train_kernel`) alike — and are
parity-green.  Written in r06 to rule the pattern out as the cause of a wrong-result experiment (csrc/qat_s16.hip i8_matvecs with seeded accumulators).
usage: tools/mfma_overlap_scan.py [file.hip ...]      (default: every csrc/*.hip, with the flags build.py uses)"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RNG = re.compile(r"^([va])\[(\d+):(\d+)\]$|^([va])(\d+)$")


def rng(op):
    m = RNG.match(op.strip())
    if not m:
        return None
    if m.group(1):
        return m.group(1), int(m.group(2)), int(m.group(3))
    return m.group(4), int(m.group(5)), int(m.group(5))


def overlap(a, b):
    return a and b and a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2]


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
    bad, ab, n, kernel = [], 0, 0, "?"
    for ln in text.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
        t = ln.strip()
        if not t.startswith("v_mfma"):
            continue
        n += 1
        ops = [o.strip() for o in t.split(";")[0].split(None, 1)[1].split(",")]
        d, a, b, c = (rng(o) for o in ops[:4])
        if overlap(d, c) and d != c:
            bad.append((kernel, t))
        if overlap(d, a) or overlap(d, b):
            ab += 1
    return os.path.basename(src), n, ab, bad


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "opendpd_amd", "csrc", "*.hip")))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        res = list(pool.map(scan, files))
    rc = 0
    for name, n, ab, bad in res:
        print(f"{name}: {n} v_mfma, {ab} with vDst over an A / B operand, {len(bad)} with vDst partly over SrcC")
        for k, t in bad[:10]:
            print(f"   {k[:70]}: {t}")
            rc = 0      # (information only)
    return rc


if __name__ == "__main__":
    sys.exit(main())
