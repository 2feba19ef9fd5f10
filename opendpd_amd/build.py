"""Compiles the HIP sources under opendpd_amd/csrc into opendpd_amd/lib/libopendpd_hip.so (gfx950)."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libopendpd_hip.so")
ARCH = "gfx950"


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


def build(force=False, verbose=False, extra_flags=(), jobs=None):
    """hipcc --offload-arch=gfx950 -fPIC -c per source (in parallel), then one -shared link; returns the library path."""
    if not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(_HERE, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    # -fno-slp-vectorize: v_pk_{fma,mul,add}_f32 issue at half rate on gfx950 (profiles/r01/ubench_issue_costs.md),
    # so SLP-formed packed math only costs v_mov shuffles and registers in these kernels
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", *extra_flags]
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    jobs = jobs or max(1, min(8, os.cpu_count() or 1))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(compile_one, sources()))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
