"""Builds libht_mi355x.so (hand-written HIP kernels + the C-ABI) in-tree with hipcc for gfx950.

    python -m hand_tracking_samples_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps the solver's fp32 evaluation order identical to the
reference CPU path (no FMA contraction); divisions and square roots stay correctly rounded (HIP default).
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("HT_LIB_PATH") or os.path.join(HERE, "libht_mi355x.so")      # HT_LIB_PATH: measurement only (A/B of two builds on one device)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
# k_solve runs one wave per SIMD, where every issued instruction (an s_nop covering a DPP hazard included) costs its 4-cycle slot: for that
# file the ILP-first scheduler fills hazard slots with independent work (58 -> 46 no-ops per pair of linear steps, 11 -> 3 per pair of
# chain rows; same arithmetic, another order of independent instructions).  Measured per file; it slows the other kernels down.
# -fno-slp-vectorize for the same file: packing the two multiplies of a cross product into v_pk_mul_f32 takes their DPP operands away (packed
# instructions cannot carry one), which costs two v_mov_b32_dpp and register shuffles per row: 42.5 -> 38.75 issued instructions per chain row, 179 -> 112 VGPRs.
# -Wno-pass-failed: k_solve is compiled for two waves per SIMD (the register budget of its eight-frames-per-CU build); the builds whose LDS footprint allows one wave per
# SIMD anyway would each warn that the target was missed.
FILE_FLAGS = {"ht_solver.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fno-slp-vectorize", "-Wno-pass-failed"]}
if os.environ.get("HT_SOLVER_FLAGS") is not None:      # measurement builds: try other per-file flags for the solver
    FILE_FLAGS = dict(FILE_FLAGS, **{"ht_solver.hip": os.environ["HT_SOLVER_FLAGS"].split()})
OBJDIR = os.path.join(HERE, "build_tuning" if os.environ.get("HT_TUNING") else "build")      # a measurement build keeps its objects apart from the product's
if os.environ.get("HT_EXTRA_FLAGS"):      # measurement builds only (with HT_LIB_PATH): e.g. -DHT_EPA_INLINE=__noinline__ for an A/B of the contact kernel
    FLAGS = FLAGS + os.environ["HT_EXTRA_FLAGS"].split()
    OBJDIR = os.path.join(HERE, "build_alt")
if os.environ.get("HT_TUNING"):      # measurement builds only: lets HT_DEBUG_SKIP / HT_NO_SIDE / HT_NO_OVERLAP reach the kernels (tools/ablate_*.sh, tools/solve_stats.py)
    FLAGS = FLAGS + ["-DHT_TUNING"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def stale():
    if os.environ.get("HT_LIB_PATH"):
        return not os.path.exists(LIB)
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "ht_mi355x.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJDIR, exist_ok=True)
    jobs = []
    for src in sources():
        name = os.path.basename(src)
        obj = os.path.join(OBJDIR, name[:-4] + ".o")
        jobs.append(([HIPCC] + FLAGS + FILE_FLAGS.get(name, []) + ["-c", src, "-o", obj], obj))

    def run(job):
        # the compiler's per-kernel resource remarks (registers, scratch, LDS) go to build/<file>.usage.txt: resource_usage() below, tests/test_build_resources.py
        cmd = job[0] + ["-Rpass-analysis=kernel-resource-usage"]
        if verbose:
            print(" ".join(job[0]), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        remarks = [l for l in r.stderr.splitlines() if "remark:" in l]
        other = [l for l in r.stderr.splitlines() if "remark:" not in l and not re.match(r"\s*(\d+ \| |\| )", l)]      # without the source lines quoted under the remarks
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise subprocess.CalledProcessError(r.returncode, cmd)
        if other and verbose:
            sys.stderr.write("\n".join(other) + "\n")
        with open(job[1][:-2] + ".usage.txt", "w") as f:
            f.write("\n".join(remarks) + "\n")
        return job[1]

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(run, jobs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-ldl", "-o", LIB]      # -ldl: RCCL is opened at run time, only by ht_comm_init (csrc/ht_comm.hip)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def resource_usage():
    """{kernel name (mangled): {"VGPRs": n, "AGPRs": n, "ScratchSize": bytes per lane, "LDS": bytes, "Occupancy": waves per SIMD, ...}} of the last build."""
    out = {}
    for f in sorted(os.listdir(OBJDIR)) if os.path.isdir(OBJDIR) else []:
        if not f.endswith(".usage.txt"):
            continue
        cur = None
        for line in open(os.path.join(OBJDIR, f)):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {})
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).strip().split()[0] if m.group(1).strip() not in ("SGPRs Spill", "VGPRs Spill", "LDS Size") else m.group(1).strip()] = int(m.group(2))
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
