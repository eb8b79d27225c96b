"""Builds libkaamer_hip.so (gfx950) in-tree with hipcc.

    python -m kaamer_amd.build [--force]

The library is the product: HIP kernels + C-ABI (include/kaamer_hip.h).  hipcc
cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkaamer_hip.so")
SOURCES = ["search.hip", "builder_device.hip", "align.hip", "builder.cpp", "host_search.cpp", "makedb.cpp"]
# every header / included kernel file: an edit to any of them rebuilds the library
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))) + [os.path.join("..", "..", "include", "kaamer_hip.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function"] + os.environ.get("KAAMER_EXTRA_CFLAGS", "").split() + \
          ["-o", LIB + ".tmp"] + srcs + ["-lpthread", "-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
