"""Build recipe for libb2h.so (hipcc, gfx950 only, in-tree).

    python -m hand_pose_sl_amd.build [--force]

The library is a plain C-ABI shared object (include/b2h.h); it links only the
HIP runtime.  hipcc cross-compiles without a GPU, so this runs in the build
container; the resulting .so travels to the GPU box with the source tree.
"""
import fcntl
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libb2h.so")
SOURCES = ["b2h_api.hip"]
HEADERS = [os.path.join("dev", "b2h_dev.h"), os.path.join("dev", "b2h_dev_exports.h"), "b2h_common.h", "kernel_mfma.h", "kernel_mfma16.h", "kernel_mfma16w.h", "kernel_mfma3.h", "kernel_mfma3w.h", "kernel_tenc.h", "kernel_valu.h", os.path.join("..", "..", "include", "b2h.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile libb2h.so if missing or older than its sources.  Returns its path.

    Safe under concurrent callers (one rank per GPU all importing the package at once):
    an exclusive file lock serialises the check-and-compile, the library is written to a
    temporary name and renamed into place atomically."""
    if not force and not stale():
        return LIB
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():      # another process built it while we waited
                return LIB
            tmp = f"{LIB}.tmp.{os.getpid()}"
            cmd = [_hipcc(), "-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-shared",
                   "-Wall", "-Wno-unused-function", "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
            if os.environ.get("B2H_ABLATE"):  # development: timing-only ablation builds (kernel_mfma16.h)
                cmd.insert(1, "-DB2H_ABLATE=" + os.environ["B2H_ABLATE"])
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
