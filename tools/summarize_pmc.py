#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel stats + counter_collection) per kernel name.
usage: tools/summarize_pmc.py <dir> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "train_kernel"
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    print("==", os.path.relpath(f, d))
    for r in csv.DictReader(open(f)):
        if "odpd" in r["Name"]:
            print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']}%")
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.relpath(f, d))
    for k, cs in acc.items():
        print("  ", k)
        for c, v in sorted(cs.items()):
            print(f"      {c:32s} mean/dispatch {sum(v)/len(v):16.1f}   (n={len(v)})")
