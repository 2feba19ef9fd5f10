#!/usr/bin/env python3
"""One character per instruction of a kernel's large basic blocks: M mfma, v valu, t transcendental, L lds, G global/scratch, n s_nop (digit = wait
states), w s_waitcnt, s other scalar.   usage: tools/isa_stream.py file.s <mangled-substring> [min_block_size]"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 100
name = [m for m in re.findall(r"^(_Z\w+):", s, re.M) if key in m][0]
a = s.index(name + ":")
b = s.index(".Lfunc_end", a)
cur, blocks = None, []
for ln in s[a:b].split("\n"):
    m = re.match(r"^(\.LBB\d+_\d+):", ln)
    t = ln.strip()
    if m:
        cur = [m.group(1), []]
        blocks.append(cur)
    elif cur is not None and t and not t.startswith(";") and not t.startswith("."):
        cur[1].append(t)


def ch(i):
    op = i.split()[0]
    if op.startswith("v_mfma"):
        return "M"
    if op in ("v_exp_f32_e32", "v_rcp_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e32", "v_log_f32_e32"):
        return "t"
    if op.startswith("v_permlane"):
        return "p"
    if op.startswith("v_"):
        return "v"
    if op.startswith("ds_"):
        return "L"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "G"
    if op == "s_nop":
        return str(min(9, int(i.split()[1]) + 1))
    if op == "s_waitcnt":
        return "w"
    return "s"


for lab, ins in blocks:
    if len(ins) >= minsz:
        st = "".join(ch(i) for i in ins)
        print(lab, len(ins))
        for k in range(0, len(st), 120):
            print("   ", st[k:k + 120])
