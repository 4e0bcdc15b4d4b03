"""Builds libsrfdet3d_hip.so (hipcc, gfx950 only) in-tree.  `python -m srfdet3d_amd.build [--force]`."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsrfdet3d_hip.so")
DEV_LIB = os.path.join(HERE, "libsrfdet3d_hip_dev.so")   # never the production path
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math",
         "-ffp-contract=off",  # kernels write their fma chains explicitly (parity with the oracle)
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "srfdet3d.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, dev=False):
    """dev=True (`--dev`): -DSRF_DEV, the developer build with the timing-ablation kernels (wrong outputs by design), the stamp
    hooks and their knobs (SRF_WINO_DBG); written to libsrfdet3d_hip_dev.so (own object directory), never to the path of the
    library that ships and that the tests and bench.py load; `_lib.py` loads it only under SRF_DEV_LIB=1 and checks
    srf_build_flavour() either way."""
    if not force and not dev and not stale():
        return LIB
    objs = []
    procs = []
    objdir = os.path.join(HERE, "build", "dev" if dev else "prod")
    os.makedirs(objdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + (["-DSRF_DEV=1"] if dev else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    out = DEV_LIB if dev else LIB
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv or "--dev" in sys.argv, verbose=True, dev="--dev" in sys.argv))
