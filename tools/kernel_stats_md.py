#!/usr/bin/env python3
"""rocprofv3 *_kernel_stats.csv -> markdown table of the odpd:: kernels.
usage: tools/kernel_stats_md.py kernel_stats.csv "title line" > out.md"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "odpd::" in r["Name"]]
rows.sort(key=lambda r: -float(r["Percentage"]))
print(f"# {sys.argv[2]}\n")
print("Per kernel: calls, average / min / max duration (us), share of total GPU time of the run.  Row-rotated kernels (`gru_*`, "
      "`lstm_*`, `delta_*`, `janet_*`, `qgru_*`) and the gate-parallel train kernels (`*_gp_train_kernel`) serve B = 256, the S16 kernels (`gru16*`, `lstm16*`, `delta16*`, `janet16*`) "
      "B = 32768; `tcnn_*<R>` both.\n")
print("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|")
for r in rows:
    print(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
          f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
