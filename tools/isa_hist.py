#!/usr/bin/env python3
"""Instruction histogram of the large basic blocks of one kernel.
usage: tools/isa_hist.py file.hip <mangled-substring> [min_block_size]"""
import collections
import re
import subprocess
import sys

src, key = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 60
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", "/tmp/_isa.s"] + [a for a in sys.argv[4:]],
                      stderr=subprocess.DEVNULL)
s = open("/tmp/_isa.s").read()
names = [m for m in re.findall(r"^(_Z\w+):", s, re.M) if key in m]
name = names[0]
a = s.index(name + ":")
b = s.index(".Lfunc_end", a)
blocks, cur = [], None
for ln in s[a:b].split("\n"):
    m = re.match(r"^(\.LBB\d+_\d+):", ln)
    t = ln.strip()
    if m:
        cur = [m.group(1), []]
        blocks.append(cur)
    elif cur is not None and t and not t.startswith(";") and not t.startswith("."):
        cur[1].append(t)
print(name, "total instrs", sum(len(b[1]) for b in blocks))
for lab, ins in blocks:
    if len(ins) >= minsz:
        c = collections.Counter(i.split()[0] for i in ins)
        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        print(f"\n{lab}: {len(ins)} instrs, VALU {valu}, MFMA {sum(v for k, v in c.items() if k.startswith('v_mfma'))}, "
              f"DS {sum(v for k, v in c.items() if k.startswith('ds_'))}, SALU/other {sum(v for k, v in c.items() if k.startswith('s_'))}")
        print("  " + ", ".join(f"{k}:{v}" for k, v in sorted(c.items(), key=lambda x: -x[1])[:28]))
