#!/usr/bin/env python3
"""Issue-class census of a kernel's large basic blocks (its unrolled step / block bodies): how many vector instructions of the FAST class (all-VGPR /
literal v_mul, v_add, v_sub, v_fmac, v_fma, v_and, v_or, v_mov: ~1.1 ns per wave64 instruction per SIMD at >= 2 waves per SIMD on an MI355X), how many
of the same opcodes with a scalar operand, how many of the FOUR-CYCLE class (everything else: conversions, v_med3, v_max, v_rndne, shifts, v_perm,
compares, v_cndmask, DPP forms; ~1.8 ns), transcendentals (3.45 ns), MFMAs by form, LDS and scratch operations — and the time those counts add up to with
the costs of profiles/r06/ubench_issue_costs.txt (f32 16x16x4 MFMA 13.6 ns, bf16 / i8 16x16x32 7.4 ns).  A lower bound of the block's time on one SIMD
when nothing overlaps; compare with the measured time per block.
usage: tools/issue_class_census.py file.hip <mangled-substring> [min block size] [extra hipcc flags ...]"""
import collections
import re
import subprocess
import sys

FAST = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_fma_f32", "v_fmaak_f32", "v_fmamk_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32")
TRANS = ("v_exp_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32")
NS = {"fast": 1.1, "fast opcode, scalar operand": 1.77, "four-cycle": 1.8, "transcendental": 3.45, "mfma f32 16x16x4": 13.6, "mfma 16x16x32": 7.4}


def own_flags(src):
    out = []
    with open(src) as f:
        for _, ln in zip(range(60), f):
            if ln.startswith("// odpd-build-flags:"):
                out += ln.split(":", 1)[1].split()
    return out


def main():
    src, key = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only", "-S", src, "-o", "/tmp/_icc.s",
                           *own_flags(src), *sys.argv[4:]], stderr=subprocess.DEVNULL)
    s = open("/tmp/_icc.s").read()
    names = [m for m in re.findall(r"^(_Z\w+):", s, re.M) if key in m]
    a = s.index(names[0] + ":")
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
    print(f"`{names[0]}`\n")
    print("| block | instructions | fast | fast opcode, scalar operand | four-cycle | transcendental | MFMA f32 16x16x4 | MFMA 16x16x32 | LDS | scratch | issue sum ns |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for lab, ins in blocks:
        if len(ins) < minsz:
            continue
        c = collections.Counter()
        for i in ins:
            op = i.split()[0]
            if op.startswith("ds_"):
                c["lds"] += 1
            if op.startswith("scratch_"):
                c["scratch"] += 1
            if not op.startswith("v_"):
                continue
            if op.startswith("v_mfma"):
                c["mfma f32 16x16x4" if "16x16x4_f32" in op or "4x4x1" in op else "mfma 16x16x32"] += 1
                continue
            base = op.replace("_e32", "").replace("_e64", "")
            ops = i[len(op):]
            has_s = bool(re.search(r"\bs\d+|\bs\[|vcc|exec", ops))
            if base in TRANS:
                c["transcendental"] += 1
            elif base in FAST and "dpp" not in i and not has_s:
                c["fast"] += 1
            elif base in FAST and "dpp" not in i:
                c["fast opcode, scalar operand"] += 1
            else:
                c["four-cycle"] += 1
        ns = sum(c[k] * v for k, v in NS.items())
        print(f"| {lab} | {len(ins)} | {c['fast']} | {c['fast opcode, scalar operand']} | {c['four-cycle']} | {c['transcendental']} | {c['mfma f32 16x16x4']} | "
              f"{c['mfma 16x16x32']} | {c['lds']} | {c['scratch']} | {ns:.0f} |")


if __name__ == "__main__":
    main()
