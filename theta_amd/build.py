"""
Builds libtheta_hip.so (gfx950) in-tree with hipcc.  `python -m theta_amd.build [--force]`.

The shared library is the product: the Python layer loads it with ctypes and fails loudly when it
is missing.  There is no CPU fallback.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtheta_hip.so")
ARCH = "gfx950"

# batch.hip restates the reference's per-interval arithmetic: no fused multiply-add contraction.
# (source, extra flags[, object name]): n3_sieve.hip is compiled twice -- the second object is the WITNESS build of the very same
# kernel source (per-candidate records of what the sieve left a candidate at, theta_search_witness)
UNITS = [
    ("n2.hip", []),
    ("n3.hip", []),
    ("n3_enum.hip", []),
    ("n3_sieve.hip", []),
    ("n3_sieve.hip", ["-DSV_WITNESS=1", "-Wno-pass-failed"], "n3_sieve_witness.o"),
    ("bnb.hip", []),
    ("batch.hip", ["-ffp-contract=off"]),
    ("api.hip", []),
    ("comm.hip", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fgpu-rdc" if False else "-fno-gpu-rdc",
          "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp") or f.endswith(".inc")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "theta_hip.h"))
    objs = []
    cc = hipcc()
    procs = []
    for unit in UNITS:
        src, extra = unit[0], unit[1]
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, unit[2] if len(unit) > 2 else src.replace(".hip", ".o"))
        objs.append(o)
        if not force and _newer(o, [s] + headers):
            continue
        cmd = [cc] + COMMON + extra + os.environ.get("THETA_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]   # (tuning builds: -DSV_OCC=4 ...)
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((os.path.basename(o), subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or not _newer(LIB, objs):
        cmd = [cc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
