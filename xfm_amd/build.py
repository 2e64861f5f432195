"""Build libxfm_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libxfm_hip.so")
SOURCES = ["capi.hip", "gemm.hip", "layernorm.hip", "attention.hip", "attention_vit.hip", "elementwise.hip", "encoder.hip", "losses.hip", "common.h",
           os.path.join("..", "..", "include", "xfm_hip.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-Wno-inline-asm",
           os.path.join(CSRC, "capi.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
