"""Build the HIP shared library in-tree (nclone_amd/libnpp_amd.so) with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU box.
-ffp-contract=off is part of the numerical contract: the reference is CPython float arithmetic (no FMA).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libnpp_amd.so")
SOURCES = ["npp_kernels.hip", "npp_render.hip", "npp_capi.cpp", "npp_level.cpp"]
HEADERS = ["npp_internal.hpp", "npp_level.hpp", os.path.join("..", "..", "include", "npp_amd.h")]


def needs_build():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [
        hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
        "-I", os.path.join(HERE, "..", "include"), "-o", OUT,
    ] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force=True, verbose="-v" in sys.argv)
    print(OUT)
