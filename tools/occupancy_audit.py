#!/usr/bin/env python3
"""Waves per SIMD each kernel ACTUALLY gets on an MI355X CU (512 registers per SIMD lane, 160 KB of LDS, four SIMDs) against what its register
count alone would allow: a kernel compiled for four waves per SIMD whose workgroup takes 90 KB of LDS runs at two (r06: qat16_fwd_kernel,
delta16_fwd_kernel).  Joins a rocprofv3 --kernel-trace CSV (registers, workgroup, grid, time; it records only the STATIC part of the LDS) with the
`[odpd-lds]` lines the library prints under $ODPD_AUDIT_LDS=1 (dynamic LDS bytes of every launch that goes through allow_big_lds).  One line per
distinct (kernel, registers, workgroup, grid); `<-- LDS` marks launches whose LDS allocation, not their registers or their grid, limits them.
usage: tools/occupancy_audit.py <kernel_trace.csv> <stderr log with [odpd-lds] lines> [min total ms]"""
import collections
import csv
import re
import subprocess
import sys

CUS, LDS_CU, REGS = 256, 160 * 1024, 512
lds_of, by_regs_anon = collections.defaultdict(set), collections.defaultdict(set)
for ln in open(sys.argv[2], errors="replace"):
    m = re.match(r"\[odpd-lds\] (\S+) lds=(\d+) regs=(-?\d+)", ln)
    if m and m.group(1) != "?":
        lds_of[m.group(1)].add(int(m.group(2)))
    elif m:       # file-local kernels (anonymous namespace): their stubs have no dynamic symbol for dladdr — joined by register count instead
        by_regs_anon[-(-int(m.group(3)) // 8) * 8].add(int(m.group(2)))
names = list(lds_of)
if names:      # stub symbols -> the trace's demangled kernel names
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    lds_of = {d.replace("__device_stub__", ""): lds_of[n] for n, d in zip(names, dem)}
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if name.startswith("__amd") or "at::native" in name or "rocfft" in name.lower():
        continue
    # (the trace reports a wave64 kernel's register allocation halved: 128 for a kernel compiled to 256 — checked against the compiler's
    # -Rpass-analysis=kernel-resource-usage figures in opendpd_amd/lib/kernel_resources.json)
    key = (name, int(r["LDS_Block_Size"]), 2 * (int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])), int(r["Workgroup_Size_X"]), int(r["Grid_Size_X"]))
    e = rows.setdefault(key, [0, 0.0])
    e[0] += 1
    e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
print("| kernel | calls | avg ms | workgroup | grid (workgroups) | registers | LDS KB (static + dynamic) | waves/SIMD by registers | by LDS | by grid | runs at | |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for (name, lds_static, regs, wg, grid), (calls, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if ms < min_ms:
        continue
    key = name[5:] if name.startswith("void ") else name
    dyn = lds_of.get(name) or lds_of.get(key) or set()
    if not dyn and "anonymous namespace" in key:
        dyn = by_regs_anon.get(regs, set())
    wpb = max(1, wg // 64)                          # waves per workgroup
    per_simd = max(1, -(-wpb // 4))                 # waves a workgroup puts on one SIMD
    by_regs = min(8, REGS // max(8, -(-regs // 8) * 8))
    blocks_regs = by_regs // per_simd
    blocks_grid = -(-(grid // wg) // CUS)
    short = key.replace("(anonymous namespace)::", "").replace("odpd::", "")
    short = short[:short.index("(")] if "(" in short else short
    # (a kernel launched with several dynamic sizes — different batch shapes — is listed once per size: the trace does not say which launch had which)
    for lds in sorted(dyn) or [0]:
        tot = lds_static + lds
        blocks_lds = LDS_CU // tot if tot else 99
        blocks = min(blocks_regs, blocks_lds, blocks_grid)
        flag = "<-- LDS" if blocks_lds < min(blocks_regs, blocks_grid) else ""
        print(f"| `{short[:72]}` | {calls} | {ms / calls:.3f} | {wg} | {grid // wg} | {regs} | {tot / 1024:.1f}{'' if dyn else ' (?)'} | {blocks_regs * per_simd} | "
              f"{blocks_lds * per_simd if tot else '-'} | {blocks_grid * per_simd} | {blocks * per_simd} | {flag} |")
