#!/usr/bin/env python3
"""Build libaudiossl_hip.so (gfx950 only) in-tree with hipcc.

    python audio-ssl_amd/build.py [--force] [--jobs N]

Every `csrc/*.hip` is compiled to `build/<name>.o` (only when stale) and linked into
`audio-ssl_amd/lib/libaudiossl_hip.so`.  hipcc cross-compiles without a GPU.
"""
import argparse
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libaudiossl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, obj, extra=()):
    cmd = [HIPCC] + FLAGS + list(extra) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return src, r.returncode, r.stdout + r.stderr


def build(force=False, jobs=None, verbose=True, ablate=False):
    """ablate: the diagnostics build - -DAUDIOSSL_ABLATE compiles in the ablation variants of the hand-scheduled kernels (no
    MFMA / no DMA / no fragment reads ...: WRONG results by design, selected by AUDIOSSL_*_DBG) into lib/libaudiossl_hip_ablate.so;
    tools load it through AUDIOSSL_LIB_PATH.  The product library never contains them."""
    global OBJ, LIB
    flags_extra = []
    if ablate:
        OBJ, LIB = os.path.join(HERE, "build_ablate"), os.path.join(LIBDIR, "libaudiossl_hip_ablate.so")
        flags_extra = ["-DAUDIOSSL_ABLATE"]
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    todo, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            todo.append((s, o))
    jobs = jobs or min(8, os.cpu_count() or 1)
    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            for src, rc, out in ex.map(lambda a: _compile(*a, extra=flags_extra), todo):
                if verbose or rc:
                    print(f"[hipcc] {os.path.basename(src)} rc={rc}")
                if out.strip() and (rc or verbose):
                    print(out)
                if rc:
                    raise RuntimeError(f"hipcc failed on {src}")
    if todo or force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[link] {LIB}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--ablate", action="store_true", help="diagnostics library with the ablation kernel variants (wrong results by design)")
    a = ap.parse_args()
    build(a.force, a.jobs, ablate=a.ablate)
    sys.exit(0)
