"""Build libgsd.so for gfx950 in-tree (gelslim_depth_amd/csrc/libgsd.so).

hipcc cross-compiles without a GPU, so this runs in the build container and the resulting .so
travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SOURCES = ["gsd_conv3x3.hip", "gsd_conv3x3_w43.hip", "gsd_convT.hip", "gsd_wgrad.hip", "gsd_wgrad_w43.hip", "gsd_pointwise.hip", "gsd_dataset.hip", "gsd_bf16_conv.hip", "gsd_bf16_pointwise.hip", "gsd_bf16_wgrad.hip"]
OUT = os.path.join(CSRC, "libgsd.so")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES]
    deps += [os.path.join(d, f) for d in (CSRC, INCLUDE) for f in os.listdir(d) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{INCLUDE}",
           "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
