"""Builds libtpspp_hip.so (the C-ABI library of include/tpspp.h) with hipcc for gfx950.

In-tree build: the .so lands next to this file so that it travels with a `gpurun` snapshot and is
visible to the driver's "which native code was loaded" check.  hipcc cross-compiles without a GPU.

    python -m tps_pp_amd.build [--force] [--verbose]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtpspp_hip.so")
ARCH = "gfx950"

# -ffp-contract=off: hipcc's default (fast) would fuse a*b+c on its own; the parity contract
# (include/tpspp.h) allows exactly the fmaf() calls written in the sources and nothing else.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-fno-fast-math", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


# per-file extra flags: tpspp_warp_geo.hip without the SLP vectoriser (it packs the grid chains into v_pk_fma_f32: slower)
EXTRA = {"tpspp_warp_geo.hip": ["-fno-slp-vectorize"]}
if os.environ.get("TPSPP_BUILD_WIDE_LAB"):      # timing experiments of tpspp_conv3_wide.hip (scripts/debug/bench_wide.py)
    EXTRA["tpspp_conv3_wide.hip"] = ["-DTPSPP_WIDE_LAB"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(ROOT, "include", "tpspp.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per .hip file (compiled in parallel, rebuilt only when that file or a header changed), then one link."""
    if not force and not _stale():
        return LIB
    # several ranks / pytest workers may call build() at once: one builds, the others wait and re-check
    import fcntl
    os.makedirs(os.path.join(CSRC, ".build"), exist_ok=True)
    with open(os.path.join(CSRC, ".build", "lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libtpspp_hip.so")
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, ".build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(ROOT, "include", "tpspp.h"), os.path.abspath(__file__)]
    hdr_t = max(os.path.getmtime(h) for h in headers)
    cflags = [f for f in FLAGS if f != "-shared"] + ["-I", os.path.join(ROOT, "include"), "-I", CSRC]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            return obj, None
        cmd = [hipcc] + cflags + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return obj, (r.returncode, r.stdout)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(compile_one, sources()))
    for obj, res in results:
        if res is not None and res[0] != 0:
            raise RuntimeError("hipcc failed:\n" + res[1])
        if verbose and res is not None and res[1].strip():
            print(res[1])
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden"] + [o for o, _ in results] + ["-o", tmp]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True))
