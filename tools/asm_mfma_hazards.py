#!/usr/bin/env python3
"""Static check of the gfx950 ISA hipcc emits for csrc/*.hip: instruction-bearing INLINE ASM next to MFMAs on the same registers.

The compiler inserts the wait states the MFMA (XDL) pipeline needs — a VALU read of an accumulator that a v_mfma is still writing, a
v_mfma read of a register a VALU op has just written — only between instructions it can see.  An `asm("v_...")` statement is opaque
to its hazard recognizer (and to the post-RA scheduler's latency model), so such an instruction placed directly behind / in front of
a v_mfma on the same register reads / feeds a stale value.  That was the r02 wrong-result build (csrc/odpd_s16.h: relu_ as
`asm("v_max_f32")` on the fc_hid accumulator; r04 root cause) and r02's inline-asm v_cndmask in front of a v_mfma.

For every kernel the script walks each basic block, finds the `;;#ASMSTART ... ;;#ASMEND` regions and reports
  RAW  an asm instruction READS a VGPR that a v_mfma within the previous `--raw` issue slots writes (default 18: the longest XDL
       write-back distance of the f32 16x16x4 / 4-block variants used here, gfx940 hazard table)
  WAR  a v_mfma within the next `--war` issue slots (default 3) READS a VGPR the asm instruction writes
  WAR-SrcC  the asm instruction WRITES a VGPR that a v_mfma within the previous `--raw` slots reads as its C operand (the XDL pipeline
       reads SrcC over several passes: the register allocator may hand that dead-after-issue source to the asm's output — the r02 bug).
       A register that a compiler-VISIBLE VALU instruction writes between the two is not reported (the recognizer saw that write: an
       accumulator operand "+v" of an asm group always has such a definition; a pure output "=v" — relu_ was one — has none).
       Writes over an in-flight MFMA's A / B operands (read at issue) are only counted (`... A/B overwrites`).
(`s_nop N` counts as N + 1 slots).  Exit status 1 when a WAR / WAR-SrcC case is reported (RAW cases are information).

usage: tools/asm_mfma_hazards.py [--raw N] [--war N] [-D MACRO ...] file.hip [file.hip ...]"""
import argparse
import re
import subprocess
import sys
import tempfile

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def split_ops(ins):
    """(mnemonic, written registers, read registers) of one instruction line (first operand = destination for VALU / MFMA / DS reads)"""
    ins = ins.split(";")[0].strip()
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else "", set(), set()
    ops = [o.strip() for o in parts[1].split(",")]
    mn = parts[0]
    if mn.startswith(("s_", "ds_write", "ds_store", "global_store", "scratch_store", "buffer_store")):
        return mn, set(), regs(parts[1])
    w = regs(ops[0])
    r = regs(",".join(ops[1:]))
    if mn.startswith("v_fmac") or mn.startswith("v_mac") or "dpp" in mn and mn.startswith("v_fmac"):
        r |= w                                   # accumulate: the destination is also a source
    return mn, w, r


def slots(ins):
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def scan(asm_text, raw_dist, war_dist, ab_info=None):
    findings, n_asm = [], 0
    kernel, block = None, []

    def flush():
        nonlocal n_asm
        for i, (ins, in_asm) in enumerate(block):
            if not in_asm:
                continue
            mn, w, r = split_ops(ins)
            if not mn.startswith("v_"):
                continue
            n_asm += 1
            d, j = 0, i - 1
            while j >= 0 and d < raw_dist:
                pins, p_asm = block[j]
                pm, pw, _ = split_ops(pins)
                if "mfma" in pm and (pw & r):
                    findings.append((kernel, "RAW", d, pins.strip(), ins.strip()))
                    break
                d += slots(pins.strip())
                j -= 1
            d, j = 0, i - 1                      # an in-flight v_mfma still READS (SrcC is read over several passes) what the asm overwrites
            w = set(w)
            while j >= 0 and d < raw_dist and w:
                pins, p_asm = block[j]
                pm, pw, pr = split_ops(pins)
                if not p_asm and "mfma" not in pm and pm.startswith("v_"):
                    w -= pw                      # a compiler-visible VALU write in between: the recognizer has already paid the wait for it
                if "mfma" in pm and not (pw & w):
                    # SrcC (last register operand) is read over the MFMA's passes: the hazard LLVM's recognizer covers for visible VALU
                    # writes (SMFMA16x16ReadVgprVALUWarWaitStates = 7).  SrcA / SrcB are read at issue: reported as information only.
                    srcc = regs(pins.split(";")[0].split(",")[-1])
                    if srcc & w:
                        findings.append((kernel, "WAR-SrcC", d, pins.strip(), ins.strip()))
                        break
                    if (pr & w) and ab_info is not None:
                        ab_info.append((kernel, d))
                d += slots(pins.strip())
                j -= 1
            d, j = 0, i + 1
            while j < len(block) and d < war_dist:
                nins, _ = block[j]
                nm, _, nr = split_ops(nins)
                if "mfma" in nm and (nr & w):
                    findings.append((kernel, "WAR", d, ins.strip(), nins.strip()))
                    break
                d += slots(nins.strip())
                j += 1

    in_asm = False
    for ln in asm_text.split("\n"):
        t = ln.strip()
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            flush()
            kernel, block = m.group(1), []
            continue
        if re.match(r"^\.LBB\d+_\d+:", ln) or t.startswith(".Lfunc_end"):
            flush()
            block = []
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        block.append((t, in_asm))
    flush()
    return findings, n_asm


def isa_of(path, defines=()):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only", "-S", path,
                               "-o", f.name] + [f"-D{d}" for d in defines], stderr=subprocess.DEVNULL)
        return open(f.name).read()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--raw", type=int, default=18)
    ap.add_argument("--war", type=int, default=3)
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("files", nargs="+")
    args = ap.parse_args()
    total = 0
    for path in args.files:
        ab = []
        findings, n_asm = scan(isa_of(path, args.D), args.raw, args.war, ab)
        print(f"{path}: {n_asm} inline-asm VALU instruction(s), {len(findings)} next to an MFMA on their registers"
              f" ({len(ab)} A/B overwrites of an in-flight MFMA, closest {min((d for _, d in ab), default='-')} slots)", flush=True)
        seen = set()
        for k, kind, d, a, b in findings:
            key = (k, kind, a.split()[0], b.split()[0])
            if key in seen:
                continue
            seen.add(key)
            print(f"   {kind} at distance {d}: {k[:70]}\n        {a}\n        {b}")
        total += sum(1 for f in findings if f[1] != "RAW")
    # (RAW: a VALU read of an accumulator in flight — reported for information: every forward kernel of the r01..r03 builds had relu_'s asm
    # directly behind the last v_mfma of its chain and was exact in every parity test, i.e. the hardware interlocks that dependency)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
