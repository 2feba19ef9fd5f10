#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_pmc.sh (headline) and tools/profile_cascade.sh (train_dpd kernels + cfg 4) into the round's records:
profiles/<round>/cascade_kernels_b65536_pmc.md, the two kernel_stats CSVs, and the entries of profiles/pmc_traffic.json that bench.py reports as
`traffic` (keyed by workload, valid for the sha1 of the kernel sources they were measured on).
usage: tools/pmc_tables.py <headline-dir> <cascade-dir> <round>        e.g. gpurun_out/r05_headline3 gpurun_out/r05_cascade3 r05"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

head, casc, rnd = sys.argv[1:4]
out = os.path.join(ROOT, "profiles", rnd)


def stats(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    return f, {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f)) if "odpd" in r["Name"]}


def pmc(d, sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


cf, cst = stats(os.path.join(casc, "stats"))
sq, fe, wr = pmc(casc, "pmc_sq"), pmc(casc, "pmc_fetch"), pmc(casc, "pmc_write")
rows = []
for k, (calls, us) in sorted(cst.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
    if us < 50:
        continue
    s, f, w = sq.get(k, {}), fe.get(k, {}).get("FETCH_SIZE"), wr.get(k, {}).get("WRITE_SIZE")
    if not s or f is None or w is None:
        continue
    rows.append({"kernel": k, "calls": calls, "us": us, "busy": s["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (s["GRBM_GUI_ACTIVE"] / 8), "valu": s["SQ_INSTS_VALU"] / 1024,
                 "wait": s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"], "fetch_kb": f, "write_kb": w, "hbm": (2 * f + w) * 1024, "raw": (f + w) * 1024})
short = lambda k: k.replace("void ", "").split("(odpd::SeqArgs")[0].split("(float const*")[0]
with open(os.path.join(out, "cascade_kernels_b65536_pmc.md"), "w") as fo:
    fo.write(f"# train_dpd cascade kernels + cfg 4 at bench size (65 536 x 200; cfg 4: 32 768 x 200), rocprofv3 --kernel-trace --stats + separate --pmc passes "
             f"(tools/profile_cascade.sh), {rnd} final build\n# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); HBM = (2 x FETCH_SIZE + "
             "WRITE_SIZE) KB (gfx950: FETCH_SIZE tallies 128-B requests at 64 B); uncorrected = FETCH_SIZE + WRITE_SIZE\n"
             "| kernel | avg us | MFMA busy | VALU instr / SIMD | wait-inst share of wave cycles | HBM MB / launch (uncorrected) |\n|---|---|---|---|---|---|\n")
    for r in rows:
        fo.write(f"| `{short(r['kernel'])}` | {r['us']:.1f} | {100 * r['busy']:.0f} % | {r['valu'] / 1e3:.0f} k | {100 * r['wait']:.0f} % | {r['hbm'] / 1e6:.0f} ({r['raw'] / 1e6:.0f}) |\n")
shutil.copy(cf, os.path.join(out, "cascade_b65536_kernel_stats.csv"))
hf, hst = stats(os.path.join(head, "stats"))
shutil.copy(hf, os.path.join(out, "headline_kernel_stats.csv"))

find = lambda sub: [r for r in rows if sub in r["kernel"]][0]
base = ("odpd_s16.h", "odpd_device.h", "odpd_seq.h")
note = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_cascade.sh, bench.py --steps 3), per launch; fetch doubled per MI355X_MICROARCH.md "
        f"(gfx950 FETCH_SIZE tallies 128-B requests at 64 B). {rnd} final build.")


def entry(kernels, files, extra=""):
    rs = [find(k) for k in kernels]
    return {"kernels": [{"kernel": r["kernel"], "avg_us": round(r["us"], 1), "FETCH_SIZE_KB": round(r["fetch_kb"], 1), "WRITE_SIZE_KB": round(r["write_kb"], 1),
                         "hbm_bytes_per_launch": round(r["hbm"])} for r in rs],
            "hbm_bytes_per_launch_raw": round(sum(r["raw"] for r in rs)), "hbm_bytes_per_launch": round(sum(r["hbm"] for r in rs)),
            "source_files": list(files), "source_sha1": bench.kernel_source_sha1(tuple(files)), "note": note + extra}


p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(p))
hk = [k for k in hst if "gru16_train_kernel" in k][0]
hfe, hwr = pmc(head, "pmc_fetch")[hk]["FETCH_SIZE"], pmc(head, "pmc_write")[hk]["WRITE_SIZE"]
e = d["dgru_h13_b65536_t200"]
e.update({"FETCH_SIZE_KB": round(hfe, 1), "WRITE_SIZE_KB": round(hwr, 1), "hbm_bytes_per_launch_raw": (hfe + hwr) * 1024, "hbm_bytes_per_launch": (2 * hfe + hwr) * 1024,
          "source_files": list(bench.HEADLINE_SOURCES), "source_sha1": bench.kernel_source_sha1(bench.HEADLINE_SOURCES), "avg_us": round(hst[hk][1], 1), "calls": hst[hk][0]})
d["vdlstm_h13_b32768_t200"] = entry(["lstm16_train_kernel"], ("lstm_s16.hip",) + base, " BPTT checkpoints (h, c) every two steps; algorithmic bytes 16 B x 32768 x 200 = 104.9 MB.")
if any("gru16x_train_kernel" in r["kernel"] for r in rows):      # (r06: train_pa DGRU H23 at 32 768 x 200 on the bf16 matrix pipe)
    d["dgru_h23_b32768_t200"] = entry(["gru16x_train_kernel"], ("gru_s16x.hip",) + base, " two-step h checkpoints of 24 units (float4 + float2 per lane); algorithmic bytes 104.9 MB.")
d["train_dpd_dgru13_dgru23_b65536_t200"] = entry(["gru16_fwd_kernel", "gru16x_lossdx_kernel", "gru16_bwd_kernel"], ("gru_s16.hip", "gru_s16x.hip") + base,
                                                 " The frozen-PA kernel's share = its two-step h checkpoints (839 MB written, read back once) + u, target, dL/du.")
d["train_dpd_tres15_dgru23_b65536_t200"] = entry(["delta16_fwd_kernel", "gru16x_lossdx_kernel", "delta16_bwd_kernel", "tres_skip_wgrad_kernel"], ("delta_s16.hip", "gru_s16x.hip") + base,
                                                 " delta16_bwd_kernel stages x and dL/dy in 16-step chunks (8 B per lane): an access width the guide calls uncalibrated — its UNcorrected "
                                                 "fetch already matches the bytes the kernel addresses (629 MB of checkpoints + 105 MB x + 105 MB dL/dy), so the doubled figure is an upper bound.")
d["train_dpd_qgru10_dgru23_b65536_t200"] = entry(["qat16u_fwd_kernel", "gru16x_lossdx_kernel", "qat16u_bwd_kernel"], ("qat_s16.hip", "gru_s16x.hip", "odpd_qat.h") + base)
json.dump(d, open(p, "w"), indent=1)
print(open(os.path.join(out, "cascade_kernels_b65536_pmc.md")).read())
print("headline:", hk[:60], hst[hk], "traffic MB", (2 * hfe + hwr) * 1024 / 1e6)
for k, v in d.items():
    print(k, v.get("hbm_bytes_per_launch"), v.get("source_sha1") == bench.kernel_source_sha1(tuple(v.get("source_files", bench.HEADLINE_SOURCES))))
