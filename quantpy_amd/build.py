"""Build libqtomo.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIB_DIR, "libqtomo.so")


# max-ilp: hipcc's default scheduler optimises for occupancy; these kernels run one to five waves per SIMD
# with registers to spare, and the ILP strategy (loads hoisted, independent chains interleaved) measured
# 4-5 % faster on every n = 3 kernel (profiles/round1_v11_*).
SCHED_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _source_hash():
    """SHA-256 over the HIP sources, the public header and the compiler flags."""
    import hashlib

    h = hashlib.sha256(" ".join(SCHED_FLAGS).encode())
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(os.path.dirname(_HERE), "include", "qtomo.h")]
    for path in srcs:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale():
    """True unless lib/libqtomo.so was built from exactly the sources in the tree (hash, not mtime: a shipped
    library newer than edited-and-reverted or freshly checked-out sources must not be trusted on its date)."""
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".srchash"):
        return True
    with open(LIB + ".srchash") as fh:
        return fh.read().strip() != _source_hash()


def build_profile_library(verbose=False):
    """Development aid: the same sources with -DQT_PHASE_TIMING -> lib/libqtomo_prof.so (phase stamps
    inside the kernels; loaded only when QTOMO_LIB points at it, see scripts/phase_timing.py)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    out = os.path.join(LIB_DIR, "libqtomo_prof.so")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DQT_PHASE_TIMING", *SCHED_FLAGS,
           "-o", out, os.path.join(CSRC, "qtomo.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared -fPIC csrc/qtomo.hip -> lib/libqtomo.so"""
    if not force and not _stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", *SCHED_FLAGS,
           "-o", LIB + ".tmp", os.path.join(CSRC, "qtomo.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    with open(LIB + ".srchash", "w") as fh:
        fh.write(_source_hash() + "\n")
    return LIB
