"""Compiles the HIP sources under opendpd_amd/csrc into opendpd_amd/lib/libopendpd_hip.so (gfx950)."""
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libopendpd_hip.so")
RESOURCES = os.path.join(LIBDIR, "kernel_resources.json")     # per-kernel registers / scratch of the last build
ARCH = "gfx950"

# kernels that are KNOWN to use scratch memory (register spills), with the bytes per lane they had when they were validated: a build in
# which any other kernel spills, or one of these spills more, is reported (build() prints a warning; tests/test_build_and_abi.py fails).
# A spilling instantiation is where a compiler bump can silently produce a different — once: a wrong — schedule.
def _known_scratch():
    p = os.path.join(CSRC, "known_scratch.json")
    return json.load(open(p)) if os.path.exists(p) else {}


KNOWN_SCRATCH = _known_scratch()


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built on this machine")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(_HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


_REMARK = re.compile(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): +(\S+)")


def _parse_resources(text):
    """`-Rpass-analysis=kernel-resource-usage` remarks -> {kernel: {vgprs, agprs, scratch, occupancy}}."""
    out, cur = {}, None
    for key, val in _REMARK.findall(text):
        if key == "Function Name":
            cur = out.setdefault(val, {})
        elif cur is not None and val.isdigit():
            cur[{"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy"}[key]] = int(val)
    return out


def unexpected_scratch(resources=None):
    """Kernels of the last build that use scratch memory beyond what KNOWN_SCRATCH records: [(kernel, bytes per lane, allowed)]."""
    if resources is None:
        if not os.path.exists(RESOURCES):
            return []
        resources = json.load(open(RESOURCES))
    return sorted((k, v.get("scratch", 0), KNOWN_SCRATCH.get(k, 0)) for k, v in resources.items() if v.get("scratch", 0) > KNOWN_SCRATCH.get(k, 0))


def build(force=False, verbose=False, extra_flags=(), jobs=None, out=None):
    """hipcc --offload-arch=gfx950 -fPIC -c per source (in parallel; a source whose object is newer than everything it includes is
    not recompiled unless force=True), then one -shared link; returns the library path.
    `extra_flags` / `out`: experiment builds (tools/exp_time.py) get their own object directory (keyed by the flags) and library path,
    so that they neither race with nor overwrite the in-tree build."""
    lib = out or LIB
    if not force and not extra_flags and out is None and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    tag = hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:10] if extra_flags else "default"
    if extra_flags and out is None:
        raise ValueError("a build with extra flags needs its own output path (out=...): it must not replace the in-tree library")
    objdir = os.path.join(_HERE, "..", "build", "obj", tag)
    os.makedirs(objdir, exist_ok=True)
    # -fno-slp-vectorize: v_pk_{fma,mul,add}_f32 issue at half rate on gfx950 (profiles/r01/ubench_issue_costs.md),
    # so SLP-formed packed math only costs v_mov shuffles and registers in these kernels
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-Rpass-analysis=kernel-resource-usage", *extra_flags]
    hipcc = _hipcc()

    def up_to_date(obj):
        """object, its resource record and every file of its dependency list (-MD) older than it, same flags"""
        dep, res = obj + ".d", obj + ".res.json"
        if force or not all(os.path.exists(f) for f in (obj, dep, res)):
            return None
        try:
            rec = json.load(open(res))
            names = open(dep).read().replace("\\\n", " ").split(":", 1)[1].split()
            t = os.path.getmtime(obj)
            if rec.get("flags") != flags or any(os.path.getmtime(n) > t for n in names):
                return None
            return rec["resources"]
        except (OSError, ValueError, KeyError, IndexError):
            return None

    def file_flags(src):
        """flags a source asks for itself: a line `// odpd-build-flags: <flags>` among its first 60 lines (e.g. gru_s16x.hip: VGPR-form MFMAs)"""
        out_ = []
        with open(src) as f:
            for _, ln in zip(range(60), f):
                if ln.startswith("// odpd-build-flags:"):
                    out_ += ln.split(":", 1)[1].split()
        return out_

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        own = file_flags(src)
        cached = up_to_date(obj) if not own else None
        if own and not force and all(os.path.exists(f) for f in (obj, obj + ".d", obj + ".res.json")):
            try:       # (same rule as up_to_date, with the file's own flags part of the record)
                rec = json.load(open(obj + ".res.json"))
                names = open(obj + ".d").read().replace("\\\n", " ").split(":", 1)[1].split()
                if rec.get("flags") == flags + own and all(os.path.getmtime(n) <= os.path.getmtime(obj) for n in names):
                    cached = rec["resources"]
            except (OSError, ValueError, KeyError, IndexError):
                cached = None
        if cached is not None:
            return obj, cached
        cmd = [hipcc, *flags, *own, "-MD", "-MF", obj + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{p.stderr[-4000:]}")
        noise = [l for l in p.stderr.splitlines() if "remark:" not in l and "[-Rpass-analysis" not in l and l.strip()]
        if noise and verbose:
            print("\n".join(noise))
        resources = _parse_resources(p.stderr)
        json.dump({"flags": flags + own, "resources": resources}, open(obj + ".res.json", "w"))
        return obj, resources

    jobs = jobs or max(1, min(8, os.cpu_count() or 1))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        results = list(pool.map(compile_one, sources()))
    objs = [o for o, _ in results]
    resources = {}
    for _, r in results:
        resources.update(r)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-ldl", "-o", lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if out is None:
        json.dump(resources, open(RESOURCES, "w"), indent=0, sort_keys=True)
    bad = unexpected_scratch(resources)
    if bad:
        print(f"[build] WARNING: {len(bad)} kernel instantiation(s) spill registers to scratch memory beyond the recorded allowance:")
        for k, s, allowed in bad[:20]:
            print(f"    {s:5d} B/lane (allowed {allowed}): {k}")
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
