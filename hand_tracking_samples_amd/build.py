"""Builds libht_mi355x.so (hand-written HIP kernels + the C-ABI) in-tree with hipcc for gfx950.

    python -m hand_tracking_samples_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps the solver's fp32 evaluation order identical to the
reference CPU path (no FMA contraction); divisions and square roots stay correctly rounded (HIP default).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libht_mi355x.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "ht_mi355x.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    cmd = [HIPCC] + FLAGS + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
