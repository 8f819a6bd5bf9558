"""Build libvr2.so (hand-written HIP for gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvr2.so")
SOURCES = ["api.hip", "gemv.hip", "gemv_w8.hip", "gemm.hip", "gemm_tn.hip", "elementwise.hip", "attention.hip", "attention_fwd.hip", "attention_bwd.hip",
           "sample.hip", "backward.hip", "optim.hip"]


def hipcc():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared", "-Wno-unused-value",
           "-I", os.path.join(os.path.dirname(HERE), "include"), "-o", LIB] + srcs
    if verbose:
        print("[video_r2_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
