#!/usr/bin/env python3
"""Investigation of the r02 wrong-result build (DESIGN §4 / csrc/gru_s16n.hip): the fused frozen-PA step of hidden 25..32 (four K-chunks
in the last unit tile) at TWO waves per SIMD, which the launch shape avoids.  Runs tools/s16n_crosscheck.check(pbb, ph) for ph = 25..32
with each given library ($OPENDPD_HIP_LIB, built with the eight-wave launch shape is the default since r04; -DODPD_RELU_ASM rebuilds the r02 defect) in a child process and prints, per library, the
worst cascade-step error and the cases beyond tolerance.   python tools/exp_s16n8w.py lib1.so [lib2.so ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, importlib.util
sys.path.insert(0, %r)
from opendpd_amd import _lib
lib = _lib.load()
lib.odpd_set_tuning(b"s16_min_batch", 0)
spec = importlib.util.spec_from_file_location("cc", %r)
cc = importlib.util.module_from_spec(spec); spec.loader.exec_module(cc)
for pbb in ("dgru", "gru"):
    for ph in range(25, 33):
        w, bad, kinks, n = cc.check(pbb, ph)
        print(f"  {pbb} H{ph}: cascade step {w[0]:.1e}  frozen bwd {w[1]:.1e}  fused train {w[4]:.1e}  beyond tolerance: {[(b[2], b[3], b[4][0]) for b in bad]}", flush=True)
""" % (ROOT, os.path.join(ROOT, "tools", "s16n_crosscheck.py"))

for lib in sys.argv[1:] or [""]:
    env = dict(os.environ)
    if lib:
        env["OPENDPD_HIP_LIB"] = os.path.abspath(lib)
    print(os.path.basename(lib) or "in-tree", flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(r.stdout + (r.stderr[-1500:] if r.returncode else ""), flush=True)
