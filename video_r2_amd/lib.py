"""ctypes binding of libvr2.so.  Signatures are derived from include/vr2.h so the header is the single source of truth.

The product path FAILS LOUDLY when the HIP extension is missing: there is no CPU / eager fallback anywhere in
video_r2_amd (the CPU oracle under oracle/ is test infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "vr2.h")
LIBPATH = os.environ.get("VR2_LIBPATH") or os.path.join(HERE, "libvr2.so")     # (the override: A/B runs of two builds on one box)

_CT = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "uint32_t": ctypes.c_uint32, "float": ctypes.c_float,
}


class Vr2Error(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every prototype in vr2.h."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(vr2_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(ctypes.POINTER(ctypes.c_float) if re.match(r"float\s*\*", a) else
                                    (ctypes.POINTER(ctypes.c_void_p) if "**" in a else ctypes.c_void_p))
                else:
                    argtypes.append(_CT[a.split(" ")[-2] if len(a.split(" ")) > 1 else a])
        protos[name] = (ctypes.c_char_p if "char" in ret else (ctypes.c_int64 if "int64" in ret else ctypes.c_int), argtypes)
    return protos


_lib = None
_protos = None


def load():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise Vr2Error(f"{LIBPATH} is missing: build it with `python -m video_r2_amd.build` (hipcc --offload-arch=gfx950); "
                       "video_r2_amd has no CPU fallback")
    # libvr2.so needs libamdhip64.so.7.  PyTorch-ROCm bundles its own copy of that runtime (same soname): it must be the
    # ONE runtime in the process, or streams / device pointers handed over from torch belong to a different HIP context
    # ("no ROCm-capable device is detected").  Loading torch's copy first makes the dynamic linker bind ours to it.
    import torch
    rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIBPATH)
    _protos = parse_header()
    for name, (res, args) in _protos.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return x


def call(name, *args):
    """Call an int-returning entry point; raise Vr2Error with vr2_last_error() on failure."""
    lib = load()
    fn = getattr(lib, name)
    rc = fn(*[_ptr(a) for a in args])
    if rc != 0:
        raise Vr2Error(f"{name} failed ({rc}): {lib.vr2_last_error().decode()}")
    return rc


_gemm_ws = {}


def gemm_workspace(device):
    """Register (once per process and device) the fp32 scratch of the GEMMs' split launches (vr2_set_gemm_workspace).  The
    buffer is process-global like the library's pointer to it and is never freed: an Engine that went away must not leave the
    library with a pointer into memory the allocator has handed to someone else."""
    import torch
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    ws = _gemm_ws.get(key)
    if ws is None:
        ws = torch.empty(16 << 20, dtype=torch.uint8, device=dev)
        _gemm_ws[key] = ws
    call("vr2_set_gemm_workspace", ws, ws.numel())
    return ws


def version():
    return load().vr2_version()
