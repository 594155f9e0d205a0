"""ctypes binding of libaudiossl_hip.so - the only door from Python into the HIP kernels.

The prototypes are parsed from `include/audiossl_hip.h` (single source of truth for the C ABI).  There is
NO fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
PyTorch is used above this layer for device memory and streams only.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
# AUDIOSSL_LIB_PATH: diagnostics only (tools/ load the -DAUDIOSSL_ABLATE build through it)
LIB_PATH = os.environ.get("AUDIOSSL_LIB_PATH") or os.path.join(_PKG, "lib", "libaudiossl_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "audiossl_hip.h")

F32, BF16 = 0, 1
ERRORS = {-1: "AUDIOSSL_EINVAL (bad shape / null pointer / unsupported configuration)",
          -2: "AUDIOSSL_ELAUNCH (HIP launch error)",
          -3: "AUDIOSSL_EALIGN (pointer or leading dimension not 16-byte aligned)"}

_SCALARS = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
            "unsigned long long": ctypes.c_ulonglong, "long long": ctypes.c_longlong}


def parse_header(path=HEADER_PATH):
    """-> {name: [(argname, ctype), ...]} for every `int audiossl_*(...)` prototype."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(audiossl_\w+)\s*\((.*?)\)\s*;", text, flags=re.S):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            if "*" in a:
                args.append((a.split("*")[-1].strip(), ctypes.c_void_p))
            else:
                ty, nm = a.rsplit(" ", 1)
                args.append((nm, _SCALARS[ty.replace("const ", "").strip()]))
        protos[m.group(1)] = args
    return protos


PROTOS = parse_header()
_lib = None
# Optional per-call timing (bench.py): {full_name: [(start_event, end_event, args), ...]} - HIP events recorded on the
# stream the kernel is launched on (torch's current stream).
PROFILE = None
PROFILE_ALL = False   # True: time every entry point (a missing key is created on first use), not only the keys present
PROFILE_NOTE = None   # set by a caller right before call(): algorithmic work the scalar arguments cannot express (gemm_multi)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing - build it with `python audio-ssl_amd/build.py` "
                               "(the audiossl HIP path has no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, args in PROTOS.items():
            fn = getattr(_lib, name)            # AttributeError here = header / library mismatch
            fn.restype = ctypes.c_int
            fn.argtypes = [t for _, t in args]
    return _lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("audiossl HIP kernels need device tensors (no CPU fallback exists)")
        if not x.is_contiguous():
            raise RuntimeError("audiossl HIP kernels need contiguous tensors")
        return x.data_ptr()
    return int(x)


_FAST = {}            # short or full name -> (full name, bound ctypes function, per-argument "is a pointer" flags)
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def _entry(name):
    full = name if name.startswith("audiossl_") else "audiossl_" + name
    proto = PROTOS[full]
    if proto[-1][0] != "stream":
        raise TypeError(f"{full} is a host entry point; use call_host")
    e = _FAST[name] = (full, getattr(lib(), full), tuple(ty is ctypes.c_void_p for _, ty in proto[:-1]))
    return e


def call(name, *args):
    """Call `audiossl_<name>`; the trailing `stream` argument is filled with torch's current HIP stream.
    This wrapper is on the launch path of every kernel (about 270 per training step), so it is kept flat."""
    e = _FAST.get(name)
    if e is None:
        e = _entry(name)
    full, fn, is_ptr = e
    if len(args) != len(is_ptr):
        raise TypeError(f"{full} takes {len(is_ptr)} arguments (+stream), got {len(args)}")
    conv = []
    for p, v in zip(is_ptr, args):
        if p and v is not None and not isinstance(v, int):
            if not v.is_cuda:
                raise RuntimeError("audiossl HIP kernels need device tensors (no CPU fallback exists)")
            if not v.is_contiguous():
                raise RuntimeError("audiossl HIP kernels need contiguous tensors")
            v = v.data_ptr()
        conv.append(v)
    conv.append(_raw_stream(_cur_device()))
    prof = None
    if PROFILE is not None:
        prof = PROFILE.get(full)
        if prof is None and PROFILE_ALL:
            prof = PROFILE[full] = []
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = fn(*conv)
    if prof is not None:
        e1.record()
        global PROFILE_NOTE
        note = PROFILE_NOTE
        if full in _GEMM_FAMILY:                # which instantiation the dispatch picked (joins bench.py's table to rocprofv3 names)
            note = (note, last_kernel())
        # scalar arguments only (host addresses of the multi-problem pointer arrays are ints too: dropped by magnitude)
        prof.append((e0, e1, tuple(a for a in args if isinstance(a, float) or (isinstance(a, int) and abs(a) < (1 << 40))), note))
        PROFILE_NOTE = None
    if rc != 0:
        raise RuntimeError(f"{full} failed: {ERRORS.get(rc, rc)}")


_GEMM_FAMILY = {"audiossl_gemm", "audiossl_gemm_dropout", "audiossl_gemm_multi", "audiossl_gemm_multi_sgd", "audiossl_gemm_multi_barlow", "audiossl_moco_logits"}


def last_kernel():
    """Name (rocprofv3 spelling) of the kernel the most recent GEMM-family call launched."""
    buf = ctypes.create_string_buffer(128)
    rc = lib().audiossl_last_kernel(ctypes.addressof(buf), 128)
    if rc != 0:
        raise RuntimeError(f"audiossl_last_kernel failed: {ERRORS.get(rc, rc)}")
    return buf.value.decode()


def call_host(name, *args):
    """Call a host-side (CPU) entry point: no stream argument, raw host addresses / scalars only."""
    full = name if name.startswith("audiossl_") else "audiossl_" + name
    proto = PROTOS[full]
    if len(args) != len(proto):
        raise TypeError(f"{full} takes {len(proto)} arguments, got {len(args)}")
    rc = getattr(lib(), full)(*args)
    if rc != 0:
        raise RuntimeError(f"{full} failed: {ERRORS.get(rc, rc)}")


def torch_dtype(dtype):
    return torch.float32 if dtype == F32 else torch.bfloat16
