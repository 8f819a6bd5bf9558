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


OBJ = os.path.join(HERE, "csrc", "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wno-unused-value", "-Wno-inline-asm"]


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def needs_build():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    return _stale(LIB, deps + [os.path.join(os.path.dirname(HERE), "include", "vr2.h")])


def build(force=False, verbose=True, jobs=None):
    """One object per source file (compiled in parallel, re-compiled only when the file, a csrc header or vr2.h changed), then
    one link.  A full build is ~1 min of hipcc on one core; after an edit of one kernel file it is that file's seconds."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    inc = os.path.join(os.path.dirname(HERE), "include")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(inc, "vr2.h")]
    todo = []
    for src in srcs:
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + headers):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc()] + FLAGS + ["-I", inc, "-c", src, "-o", obj]
        if verbose:
            print("[video_r2_amd.build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(compile_one, todo))
    objs = [os.path.join(OBJ, os.path.basename(src)[:-4] + ".o") for src in srcs]
    cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print("[video_r2_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
