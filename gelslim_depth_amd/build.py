"""Build libgsd.so for gfx950 in-tree (gelslim_depth_amd/csrc/libgsd.so).

hipcc cross-compiles without a GPU, so this runs in the build container and the resulting .so
travels to the GPU box with the repo snapshot.

Every source is compiled to its own object (in parallel, re-compiled only when its content, a header or the flags
changed) and the objects are linked into the shared library.  What is up to date is decided by CONTENT hashes, not
file times: on a box that received a pushed snapshot a stale .so can be newer than the sources it was not built from.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SOURCES = ["gsd_conv3x3.hip", "gsd_conv3x3_w43.hip", "gsd_conv3x3_w2d.hip", "gsd_convT.hip", "gsd_wgrad.hip", "gsd_wgrad_w43.hip", "gsd_wgrad_w2d.hip", "gsd_wgrad_first.hip", "gsd_pointwise.hip", "gsd_dataset.hip", "gsd_bf16_conv.hip", "gsd_bf16_pointwise.hip", "gsd_bf16_wgrad.hip", "gsd_bf16_first.hip", "gsd_bf16_inc.hip", "gsd_bf16_c64.hip", "gsd_bf16_ctgemm.hip"]
OUT = os.path.join(CSRC, "libgsd.so")
STAMP = OUT + ".stamp"      # sha256 of everything the .so was built from (git-ignored, travels with the .so)
OBJ = os.path.join(CSRC, "obj")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _headers():
    return sorted(os.path.join(d, f) for d in (CSRC, INCLUDE) for f in os.listdir(d) if f.endswith(".h"))


def _digest(paths, extra: str = "") -> str:
    h = hashlib.sha256((" ".join(CFLAGS) + extra).encode())
    for d in paths:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def source_digest() -> str:
    """sha256 over the compile flags and the contents of every source and header, in a fixed order."""
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + _headers())


def needs_build() -> bool:
    if not (os.path.exists(OUT) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_digest()


def _compile_one(hipcc: str, src: str, headers, verbose: bool) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    stamp = obj + ".stamp"
    digest = _digest([os.path.join(CSRC, src)] + headers)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return obj
    cmd = [hipcc] + CFLAGS + [f"-I{INCLUDE}", "-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    with open(stamp, "w") as fh:
        fh.write(digest + "\n")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    digest = source_digest()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    headers = _headers()
    jobs = max(1, min(len(SOURCES), int(os.environ.get("GSD_BUILD_JOBS", str(os.cpu_count() or 1)))))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile_one(hipcc, s, headers, verbose), SOURCES))
    tmp = OUT + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    # the link leaves unresolved symbols unresolved (-shared): load the result once so that a kernel template that was
    # referenced but never instantiated fails the BUILD, not the first import on the GPU box
    import ctypes
    ctypes.CDLL(tmp, mode=os.RTLD_NOW)
    os.replace(tmp, OUT)
    with open(STAMP, "w") as fh:
        fh.write(digest + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
