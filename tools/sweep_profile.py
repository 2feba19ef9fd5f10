#!/usr/bin/env python3
"""cProfile of a lockstep sweep (where does the host time go?).   python tools/sweep_profile.py [K]"""
import cProfile
import os
import pstats
import sys

sys.argv = [sys.argv[0]] + sys.argv[1:]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
os.environ.setdefault("EXP_EPOCHS", "10")
import runpy  # noqa: E402

sys.argv = ["sweep_bench.py", "1"]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweep_bench.py"))
od, kw = ns["od"], ns["kw"]
pr = cProfile.Profile()
pr.enable()
od.train_pa_sweep(seeds=tuple(range(200, 200 + K)), **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
