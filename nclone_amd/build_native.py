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
SOURCES = ["npp_kernels.hip", "npp_render.hip", "npp_reach_kernel.hip", "npp_capi.cpp", "npp_level.cpp", "npp_reach.cpp", "npp_host.cpp"]
HEADERS = ["npp_internal.hpp", "npp_level.hpp", "npp_zoo.hpp", "npp_reach.hpp", "npp_reach_build.hpp", "npp_reach_features.hpp",
           "npp_host.hpp", "npp_zoo_layout.hpp",
           "npp_reach_tables.inc", os.path.join("..", "..", "include", "npp_amd.h")]


def needs_build():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False, out=None, extra_flags=()):
    """Compile every translation unit in parallel (npp_kernels.hip four times, -DNPP_TU=0..3: one (ZOO, MANY) family of step
    kernels each), then link.  `out` / `extra_flags` build a variant elsewhere (tools/ab_bench.py A/B runs)."""
    if out is None and not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "_obj") if out is None else out + "_obj"
    out = OUT if out is None else out
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
             "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-I", os.path.join(HERE, "..", "include")]
    flags += list(extra_flags)
    if verbose:
        flags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    # step kernels: machine LICM off.  It hoists ~25 registers of fp64 literals (and as many scalar values) out of the tick loops and
    # then spills to scratch memory to hold them; without it the G = 16 one-slot and uncapped builds need no scratch at all and run
    # as fast, and the zoo kernels gain 4 % (round 3: DESIGN.md 4.1, profiles/r03_kernel_resources.csv).  The render kernels showed
    # no difference and keep the default.
    NOLICM = ["-mllvm", "-disable-machine-licm"]
    jobs = [("npp_kernels.hip", ["-DNPP_TU=%d" % k] + NOLICM, "npp_kernels_tu%d.o" % k) for k in range(4)]
    jobs += [("npp_render.hip", [], "npp_render.o"), ("npp_capi.cpp", [], "npp_capi.o"), ("npp_level.cpp", [], "npp_level.o"),
             ("npp_reach_kernel.hip", [], "npp_reach_kernel.o"), ("npp_reach.cpp", [], "npp_reach.o"), ("npp_host.cpp", [], "npp_host.o")]
    procs = []
    for src, extra, obj in jobs:
        cmd = [hipcc] + flags + extra + ["-c", os.path.join(CSRC, src), "-o", os.path.join(objdir, obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [os.path.join(objdir, j[2]) for j in jobs])
    return out


HOST_SOURCES = ["npp_level.cpp", "npp_reach.cpp", "npp_host.cpp"]   # no HIP header anywhere below them
SANITIZED = os.path.join(HERE, "libnpp_host_asan.so")


def build_sanitized(force=False):
    """The GPU-free part of the library (level compiler, reachability table builder, host-only C entries) compiled with
    g++ -fsanitize=address,undefined into a TEST-ONLY library (the build needs its own sanitizer run; GPU
    AddressSanitizer is not available on the pool).  tests/test_host_sanitized.py loads it in a child process under
    LD_PRELOAD=libasan and runs the host checks plus a malformed-map fuzz."""
    srcs = [os.path.join(CSRC, f) for f in HOST_SOURCES]
    if not force and os.path.isfile(SANITIZED):
        t = os.path.getmtime(SANITIZED)
        if all(os.path.getmtime(os.path.join(CSRC, f)) <= t for f in HOST_SOURCES + HEADERS):
            return SANITIZED
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-I", os.path.join(HERE, "..", "include"),
           "-o", SANITIZED] + srcs
    subprocess.check_call(cmd)
    return SANITIZED


if __name__ == "__main__":
    # python -m nclone_amd.build_native [-v] [--out PATH] [-- extra hipcc flags]
    args = sys.argv[1:]
    extra = args[args.index("--") + 1:] if "--" in args else []
    out = args[args.index("--out") + 1] if "--out" in args else None
    print(build(force=True, verbose="-v" in args, out=out, extra_flags=extra))
