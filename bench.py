#!/usr/bin/env python3
"""bench.py -- synthetic depth frames/s of the MI355X hand-tracking hot path (CNN + pose solver), 1..8 GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8                       (starts its own 8 workers; the parent never touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the whole per-frame path (BASELINE.json configs[2]: depth normalise -> CNN -> decode -> FitError ->
[reset path] -> 5-step MultiStepSim -> accept -> 3 FitPointCloud passes -> poses) over one batch of independent 64x64
synthetic frames per GPU, inputs resident in HBM, every tracker re-seeded from its start pose (independent frames).
Frames are sharded contiguous per rank with no data-path collective; with N>1 the poses are all-gathered over RCCL once
per step (inside the timed region).  Scaling is weak: --frames-per-gpu (default 1024) is fixed as N grows; BASELINE
configs[3] (65536 frames over 8 GPUs) is `--gpus 8 --frames-per-gpu 8192`.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed on the
stream it runs on) and `cpu_baseline` (the reference's own code from oracle/_ref when present, else the C oracle port).
The batch is 1024 DISTINCT animation-bank frames (bench_data/frames1024.npz, row (3 + 9 i) mod 2336: SURVEY 8d config 2;
larger batches repeat them, every rank starts at another offset); after the timed loop EVERY distinct frame of rank 0's
batch is compared with what the reference itself produced for it (tests/golden/poses1024.htfx, `verified`), so a number
from a build that computes something else is reported as such.  Tuning switches of the library (HT_DEBUG_SKIP, HT_NO_SIDE, HT_NO_OVERLAP) make the run refuse.
"""
import argparse
import glob
import json
import os
import socket
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3      # dense fp32 MFMA (= fp32 vector) peak
FP32_VALU_PEAK_TF = 157.3
CNN_FLOP = {"cnn": 26.47e6,    # 2*(1440000 + 2359296 + 4718592 + 4718592), SURVEY 8(d)
            "cnn128": 2.0 * (124 * 124 * 25 * 16 + 28 * 28 * 256 * 64 + 12544 * 2048 + 2048 * 2304)}
TUNING_ENV = ("HT_DEBUG_SKIP", "HT_NO_SIDE", "HT_NO_OVERLAP", "HT_RESET_JOIN")
VERIFY_POS_TOL, VERIFY_QUAT_TOL, VERIFY_CNN_TOL = 2e-4, 2e-3, 2e-5      # the tolerances of tests/test_gpu_solver.py (whole path)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-gpu", type=int, default=1024)
    ap.add_argument("--workload", default="cnn+solver", choices=["cnn+solver", "cnn", "config5", "config5-cnn128", "config5-e2e"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--solve-tables", type=int, default=-1, help="measurement aid: ht_debug_solve_tables(mode) on the timed context (0 k_solve's own prologue, 1 every table from k_solve_prep, 2 the pose-only tables); default: the library's choice")
    ap.add_argument("--no-two-in-flight", action="store_true", help="skip the extra leg that times two batches in flight on two contexts")
    ap.add_argument("--no-host-io", action="store_true", help="skip the extra leg that times the same steps with pinned host buffers in and out")
    ap.add_argument("--always-take-cnn", action="store_true", help="cnn+solver workload with the application's always_take_cnn switch (synthetic-tracker.cpp:91): every frame accepts the CNN-driven pose; verified against tests/golden/poses1024_takecnn.htfx")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the pose gather even with one rank")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks as children of this process, which has not touched a GPU (and never does:
    no process that has initialised HIP is replaced or re-executed).  Rank 0's JSON line passes through on stdout."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def _tile(a, n, first=0):
    import numpy as np
    idx = (first + np.arange(n)) % len(a)
    return a[idx]


def _load_frames(n, first=0):
    """BASELINE configs[2]: 1024 distinct 64x64 frames (animation-bank row (3 + 9 i) mod 2336, tools/regen_goldens.sh), repeated for larger batches, from frame `first` on"""
    import numpy as np
    z = np.load(os.path.join(ROOT, "bench_data", "frames1024.npz"))
    return (_tile(z["depth"].reshape(-1, 4096), n, first).astype(np.uint16), _tile(z["cam"], n, first).astype(np.float32), _tile(z["startpose"], n, first).astype(np.float32))


def _load_frames5(n, first=0):
    """BASELINE configs[4]: 256 distinct 128x128 frames of the 26-bone hand (animation-bank row 3 + 9 i, ref_harness fullframes: tools/regen_goldens.sh), repeated"""
    import numpy as np
    z = np.load(os.path.join(ROOT, "bench_data", "frames5_256.npz"))
    return (_tile(z["depth"], n, first).astype(np.uint16), _tile(z["cam"], n, first).astype(np.float32), _tile(z["startpose"], n, first).astype(np.float32))


def _reference_poses(which, take_cnn=False):
    """(fixture arrays, where they come from, the per-frame spread of the reference's own builds or None)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import htfx
    import numpy as np
    G = os.path.join(ROOT, "tests", "golden")
    if which == "cnn+solver":
        name = "poses1024_takecnn" if take_cnn else "poses1024"
        f = htfx.load(os.path.join(G, name + ".htfx"))
        return ({"user": f["uw_pose_user"], "other": f["other_pose"], "initializing": f["flags"][:, 1].astype(np.int32)},
                "tests/golden/%s.htfx (the reference's unit of work on every frame, ref_harness poses%s)" % (name, " ... takecnn" if take_cnn else ""),
                np.load(os.path.join(G, "ref_spread1024_takecnn.npz" if take_cnn else "ref_spread1024.npz")))
    if which == "config5":
        f = htfx.load(os.path.join(G, "poses5full256.htfx"))
        return {"user": f["uw_pose_user"], "other": f["other_pose"], "initializing": f["flags"][:, 1].astype(np.int32)}, "tests/golden/poses5full256.htfx (the reference's HandTracker on the 256 128x128 frames, 26 bones, ref_harness posesfull)", np.load(os.path.join(G, "ref_spread5_256.npz"))
    f = htfx.load(os.path.join(G, "e2e128_256.htfx"))
    return {"user": f["all/uw_pose_user"], "other": f["all/other_pose"], "initializing": f["all/flags"][:, 1].astype(np.int32)}, "tests/golden/e2e128_256.htfx all/ (the reference's stage functions and layer classes on the 256 frames, ref_harness e2e128)", np.load(os.path.join(G, "ref_spread5e2e_256.npz"))


def verify_poses(got, other, initializing, ref, idx, against, spread, take_cnn=False, ill_conditioned=False):
    """Every distinct frame of the timed batch against the reference's result for it: the user poses AND othermodel (the CNN-driven half of the step) AND the
    tracker's `initializing` flag.  With the per-frame spread of the reference's own FMA builds at hand (tests/golden/ref_spread*.npz: every workload since round 5) the rule is tests/parity_rule.py, the one
    tests/test_gpu_batch_parity.py asserts: finite; a frame outside 2e-5 m / 2e-4 only where the reference's own builds are, by at most twice their move; nothing
    beyond 5e-3 m / 5e-2.  CNN-driven poses and configs[4]'s model: by distribution (parity_rule.distribution)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_rule as pr
    first = {}
    for slot, i in enumerate(idx):
        first.setdefault(int(i), slot)
    fr = np.array(sorted(first)); sl = np.array([first[i] for i in fr])
    out = {"against": against, "frames_compared": int(len(fr)), "tol": [VERIFY_POS_TOL, VERIFY_QUAT_TOL]}
    if spread is not None:
        def distribution(dev, key, cap=pr.CAP_TAKE_CNN, max_factor=None):      # CNN-driven poses / an ill-conditioned model: by number and size, not by name (tests/parity_rule.py: distribution)
            sp, sq = pr.spread_of(spread, key, fr)
            return pr.distribution(dev[sl], ref[key][fr], sp, sq, cap, max_factor)
        if take_cnn:
            uok, ud = distribution(got, "user")
            out.update(ud); out["rule"] = "always_take_cnn: user poses and othermodel are CNN-driven: percentiles <= 2x and no more frames outside the tight band than the reference's own FMA builds, cap 0.1 m / 1.0"
        elif ill_conditioned:
            # configs[4]'s 26-bone hand (cloned fingers in permanent contact, 15 polytope runs per frame) amplifies a rounding difference on ~10 % of its frames in EVERY build, and which
            # frames depends on the perturbation (the reference's two FMA builds disagree with each other on them): the frames outside the band are held by number and size, not by name
            uok, ud = distribution(got, "user", pr.CAP_TAKE_CNN, 2.0)
            out.update(ud); out["rule"] = "configs[4] (ill-conditioned model): user poses and othermodel by distribution -- percentiles <= 2x, no more frames outside 2e-5 m / 2e-4 than the reference's own FMA builds, the largest <= 2x their largest"
        else:
            u = pr.summary(got[sl], ref["user"][fr], *pr.spread_of(spread, "user", fr))
            uok = u["ok"]
            out.update({k: u[k] for k in ("within_2e-5m_2e-4", "within_2e-4m_2e-3", "reference_fma_builds_within_2e-5m_2e-4", "median_abs_dpos_m", "median_abs_dquat", "max_abs_dpos_m", "max_abs_dquat", "worst_frame", "frames_failing_the_rule")})
            out["rule"] = "user poses: tests/parity_rule.py frame by frame (outside 2e-5 m / 2e-4 only where the reference's own FMA builds are, by at most twice their move, cap 5e-3 m / 5e-2); othermodel (CNN-driven): percentiles <= 2x and no more frames outside the tight band than the reference's own FMA builds"
        ook, od = distribution(other, "other", pr.CAP_TAKE_CNN, 2.0 if ill_conditioned else None)
        out["othermodel"] = od
        flags_ok = bool(np.array_equal(initializing[sl], ref["initializing"][fr]))
        out["initializing_flags_equal"] = flags_ok
        out["verified"] = bool(uok and ook and flags_ok)
        return out
    raise RuntimeError("verify_poses: no reference-spread fixture for this workload (tests/golden/ref_spread*.npz)")


def _golden8():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import htfx
    return htfx.load(os.path.join(ROOT, "tests", "golden", "golden8.htfx"))


def cpu_baseline_config5(depth, cams, start, seed, gain):
    """BASELINE configs[4] on the host: the reference itself on the 26-bone model (oracle/_ref/model_hand26.json in its own schema); else the C oracle
    (pinned bit for bit on the reference's full-frame goldens, tests/test_fullframe.py) on a bounded sample, one thread."""
    import numpy as np
    mj = os.path.join(ROOT, "oracle", "_ref", "model_hand26.json")
    if os.path.exists(mj):
        try:
            r = _reference_baseline(depth, cams, start, seed, gain, h=128, w=128, model_json=mj, nsample=64, what="HandSegmentVR + update_cnn_model + 3 passes on 128x128 frames, 26 bones")
            if r:
                return r
        except Exception as e:
            sys.stderr.write("reference baseline for configs[4] unavailable (%s); timing the C oracle port\n" % e)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_lib as ol
    from hand_tracking_samples_amd import weights as W
    o = ol.Oracle(W.make_cnnb(seed, gain), model=os.path.join(ROOT, "tests", "golden", "model_hand26.htfx"))
    o.head.par.microforce = 3.0
    o.head.par.mainthreadpasses = 3
    nsample = min(64, len(depth))
    user = np.zeros((26, 7), np.float32)
    best = 1e30
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(nsample):
            o.reset(start[i])
            cam = ol.camera(cams[i], 128, 128)
            o.L.ho_update(o.h, ol.u16ptr(np.ascontiguousarray(depth[i])), C.byref(cam), ol.fptr(user))
        best = min(best, (time.perf_counter() - t0) / nsample)
    o.close()
    return {"value": 1.0 / best, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames x 2 reps (best), C oracle -O2 -ffp-contract=off: HandSegmentVR + update_cnn_model + 3 passes, 26 bones" % nsample}


def cpu_baseline_e2e128(depth, cams, start, seed, gain):
    """BASELINE configs[4] end to end on the host: the reference's own stage functions and layer classes in the order of handtrack.h:693-785 (ref_harness bench128),
    26-bone model in the reference's schema; one thread, the IEEE build the fixture comes from and the reference Makefile's -Ofast."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_harness"); mj = os.path.join(ROOT, "oracle", "_ref", "model_hand26.json")
    if not (os.path.exists(ref_bin) and os.path.exists(mj)):
        return None
    nsample = min(48, len(depth))
    env = dict(os.environ); env["HT_REF_MODEL_JSON"] = mj
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, "frames.htfx")
        _write_htfx(fn, {"depth": depth[:nsample].reshape(-1, 128, 128), "cam": cams[:nsample], "startpose": start[:nsample]})
        def one(binary):
            out = subprocess.run([binary, "bench128", fn, hex(seed), str(gain), "2"], capture_output=True, text=True, timeout=900, env=env)
            return json.loads(out.stdout.strip().splitlines()[-1])
        r = one(ref_bin)
        res = {"value": r["frame_fps"], "unit": "frames/s", "cores": 1, "kind": "reference", "cnn_only_fps": r["cnn_fps"],
               "sample": "%d frames x 2 reps (best), reference headers built -O2 -ffp-contract=off: the 128x128-input net (the reference's layer classes), CNNOutputAnalysis with camsub(cam, 8), "
                         "FitError / MultiStepSim / accept and 3 FitPointCloud passes on the 26-bone model, stages called directly (ref_harness bench128)" % nsample}
        if os.path.exists(ref_bin + "_ofast"):
            try:
                f = one(ref_bin + "_ofast")
                res["ofast"] = {"value": f["frame_fps"], "cnn_only_fps": f["cnn_fps"], "flags": "-Ofast -march=x86-64-v3 (timing only)"}
            except Exception as e:
                sys.stderr.write("-Ofast reference baseline unavailable (%s)\n" % e)
        return res


def cpu_baseline_cnn128(x, w128):
    """C oracle of the 128x128-input net (pinned bit for bit on the reference's own layer classes, tests/test_cnn128.py), one thread."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    nsample = min(48, len(x))
    best = 1e30
    for rep in range(2):
        t0 = time.perf_counter()
        ol.cnn128_eval(w128, x[:nsample])
        best = min(best, (time.perf_counter() - t0) / nsample)
    return {"value": 1.0 / best, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "%d frames x 2 reps (best), C oracle -O2 -ffp-contract=off, CNN forward 128x128 only" % nsample}


def _write_htfx(path, arrays):
    import numpy as np
    code = {np.dtype(np.float32): 0, np.dtype(np.int32): 1, np.dtype(np.uint16): 2, np.dtype(np.uint8): 3}
    with open(path, "wb") as f:
        f.write(b"HTFX0001" + struct.pack("<I", len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            raw = a.tobytes()
            f.write(name.encode()[:47].ljust(48, b"\0"))
            f.write(struct.pack("<IIIIIIQ", code[a.dtype], a.ndim, *(list(a.shape) + [1] * (4 - a.ndim)), len(raw)))
            f.write(raw + b"\0" * ((8 - (len(raw) & 7)) & 7))


def _host_cores():
    """Cores this process may use: the cgroup's CPU quota when there is one (a GPU box hands out a share of the host), else the affinity mask; the
    all-cores leg starts one process per core, so the count is capped at 64 to keep the default run within minutes."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(float(quota) / float(period)))))
    except Exception:
        pass
    return max(1, min(n, 64))


def _reference_baseline(depth, cams, start, seed, gain, h=64, w=64, model_json=None, nsample=192, what="update_cnn_model + 3 passes"):
    """The reference's own code (oracle/_ref/ref_harness, its headers compiled where they lie) on a bounded sample of the same workload:
    one thread (IEEE build = the build all fixtures come from, and the -Ofast build of the reference's own Makefile), then every host core
    with one process per core over the same sample.  None when the harness binary is not there."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    fast_bin = ref_bin + "_ofast"
    if not os.path.exists(ref_bin):
        return None
    nsample = min(nsample, len(depth))
    env = dict(os.environ)
    if model_json:
        env["HT_REF_MODEL_JSON"] = model_json
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, "frames.htfx")
        _write_htfx(fn, {"depth": depth[:nsample].reshape(-1, h, w), "cam": cams[:nsample], "startpose": start[:nsample]})

        def one(binary, reps):
            out = subprocess.run([binary, "bench", fn, hex(seed), str(gain), str(reps)], capture_output=True, text=True, timeout=900, env=env)
            return json.loads(out.stdout.strip().splitlines()[-1])
        r = one(ref_bin, 3)
        res = {"value": r["frame_fps"], "unit": "frames/s", "cores": 1, "kind": "reference",
               "sample": "%d frames x 3 reps (best), reference headers built -O2 -ffp-contract=off (the IEEE build every fixture comes from), %s" % (nsample, what)}
        if r.get("cnn_fps") and (h, w) == (64, 64):
            res["cnn_only_fps"] = r["cnn_fps"]
        if os.path.exists(fast_bin):
            try:
                f = one(fast_bin, 3)
                res["ofast"] = {"value": f["frame_fps"], "cnn_only_fps": f.get("cnn_fps") if (h, w) == (64, 64) else None, "flags": "-Ofast -march=x86-64-v3 (the reference Makefile's -Ofast; its results differ by millimetres, SURVEY F7: timing only)"}
            except Exception as e:
                sys.stderr.write("-Ofast reference baseline unavailable (%s)\n" % e)
        # every host core: one process per core, each over the whole sample once (frames are independent, so this is how the reference would be scaled out)
        ncpu = _host_cores()
        try:
            t0 = time.perf_counter()
            procs = [subprocess.Popen([ref_bin, "bench", fn, hex(seed), str(gain), "1"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(ncpu)]
            outs = [p.communicate(timeout=900)[0] for p in procs]
            wall = time.perf_counter() - t0
            per = [json.loads(o.strip().splitlines()[-1]) for o in outs]
            # the processes' own unit-of-work clocks (start-up and the CNN-only loop excluded), all running at once
            res["all_cores"] = {"value": round(sum(q["frame_fps"] for q in per), 2), "cores": ncpu, "wall_s": round(wall, 2),
                                "sample": "%d processes x %d frames at once, sum of the processes' frames/s" % (ncpu, nsample)}
        except Exception as e:
            sys.stderr.write("all-cores reference baseline unavailable (%s)\n" % e)
        return res


def cpu_baseline(depth, cams, start, seed, gain):
    """Reference CPU path on a bounded sample of the same workload."""
    import numpy as np
    try:
        r = _reference_baseline(depth, cams, start, seed, gain)
        if r:
            return r
    except Exception as e:      # fall through to the port
        sys.stderr.write("reference baseline unavailable (%s); timing the C oracle port\n" % e)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_lib as ol
    from hand_tracking_samples_amd import weights as W
    o = ol.Oracle(W.make_cnnb(seed, gain))
    o.head.par.microforce = 3.0
    o.head.par.mainthreadpasses = 3
    nsample = min(128, len(depth))
    user = np.zeros((17, 7), np.float32)
    best = 1e30
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(nsample):
            o.reset(start[i])
            cam = ol.camera(cams[i])
            o.L.ho_update(o.h, ol.u16ptr(np.ascontiguousarray(depth[i])), C.byref(cam), ol.fptr(user))
        best = min(best, (time.perf_counter() - t0) / nsample)
    o.close()
    return {"value": 1.0 / best, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames x 2 reps (best), C oracle -O2 -ffp-contract=off, update_cnn_model + 3 passes" % nsample}


def _pmc_traffic(kernel_prefix, workload, frames_per_gpu):
    """HBM bytes per launch of a kernel from the newest committed PMC summary that was measured on this workload at this batch size, else None."""
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.json")), reverse=True):
        try:
            pmc = json.load(open(fn))
            meta = pmc.get("_measured_on", {})
            if meta.get("workload") != workload or int(meta.get("frames_per_gpu", -1)) != int(frames_per_gpu):
                continue
            vals = [v["hbm_bytes_per_launch"] for k, v in pmc.items() if k.startswith(kernel_prefix)]
            if vals:
                return max(vals)
        except Exception:
            continue
    return None


def _pmc_issue(kernel_prefix, workload, frames_per_gpu):
    """issue / L2 figures of a kernel from the newest committed counter summary (tools/pmc_solve.py) measured on this workload at this batch size, else {}"""
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_solve_issue*.json")), reverse=True):
        try:
            pmc = json.load(open(fn))
            meta = pmc.get("_measured_on", {})
            if meta.get("workload") != workload or int(meta.get("frames_per_gpu", -1)) != int(frames_per_gpu):
                continue
            best = None
            for k, v in pmc.items():
                if k.startswith(kernel_prefix) and (best is None or v.get("dispatches", 0) > best.get("dispatches", 0)):
                    best = v
            if best:
                return {"issue_frac": best.get("issue_frac"), "clocks_per_instruction": best.get("clocks_per_instruction"), "wait_any_frac": best.get("wait_any_frac"),
                        "l2_gbps": best.get("l2_gbps"), "l2_hit_rate": best.get("l2_hit_rate"), "counters_from": os.path.relpath(fn, ROOT)}
        except Exception:
            continue
    return {}


def main():
    args = parse_args()
    bad = [k for k in TUNING_ENV if os.environ.get(k)]
    if bad and os.environ.get("HT_BENCH_TUNING_RUN") != "1":      # tools/ablate_*.sh set this on a -DHT_TUNING build; such a line is not a result
        raise SystemExit("bench.py refuses to run with the library's tuning switches set (%s): the kernels would skip work" % ", ".join(bad))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # The context drives three HIP streams (main + two side streams) beside torch's and, with N > 1, RCCL's: more than the four hardware queues the HIP
    # runtime maps streams onto by default, and streams that share a queue run one after the other.  Measured on one rank (tools/dist_probe.py): with
    # the process group up a step takes 8.9 ms on 4 queues and 8.0 on 6 or 8 (8.0 - 8.1 without a process group on any count).  Must be set before the
    # runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")

    # stdout carries exactly one JSON line: whatever libraries print there (RCCL's version banner does) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from hand_tracking_samples_amd import native, weights as W
    from hand_tracking_samples_amd.shard import PoseBuffers, gather_poses, negotiate_library_gather, rank_frames

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE\n" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    use_dist = world > 1 or args.force_dist      # --force-dist: rehearse the RCCL code path with a single rank on a 1-GPU box
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    B = args.frames_per_gpu
    seed, gain = W.DEFAULT_SEED, W.DEFAULT_FC2_GAIN
    wl = args.workload
    e2e = wl == "config5-e2e"
    cfg5 = wl in ("config5", "config5-e2e")
    cnn_only = wl in ("cnn", "config5-cnn128")
    cnn128 = wl == "config5-cnn128"
    frames5 = cfg5 or cnn128

    # this rank's contiguous shard of the global frame list; the list walks the distinct frames round and round, every rank's shard starting 131 frames further on
    # ("same generators, different animbank offsets"), so ranks do not work on identical batches (hand_tracking_samples_amd/shard.py)
    ndistinct = 256 if frames5 else 1024
    frame_idx = rank_frames(B, rank, world, ndistinct)      # which distinct frame every slot carries
    depth, cams, start = (_load_frames5 if frames5 else _load_frames)(len(frame_idx), int(frame_idx[0]))
    gold = None
    if rank == 0 and wl == "cnn" and B >= 8:      # slots 0..7: the frames the reference's heat-maps are committed for
        gold = _golden8()
        for f in range(8):
            depth[f] = gold["f%d/depth" % f].reshape(-1); cams[f] = gold["f%d/cam" % f]; start[f] = gold["f%d/startpose" % f]

    # the product's own baked model (hand_tracking_samples_amd/assets/: ht_model_bake of the reference's model_hand.json, tools/bake_assets.sh); the fixtures
    # under tests/golden/ are what the REFERENCE's constructor built and only check it (tests/test_model_build.py: identical bit for bit)
    ctx = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand26.htfx" if cfg5 else "model_hand17.htfx"), B, device=local)
    if not e2e:
        ctx.load_weights(W.make_cnnb(seed, gain))
    ctx.set_params(microforce=3.0, mainthreadpasses=3, **({"always_take_cnn": 1} if args.always_take_cnn else {}))        # synthetic-tracker.cpp:91-93
    if args.solve_tables >= 0:
        ctx.debug_solve_tables(args.solve_tables)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    d_cams = torch.from_numpy(cams).to(dev)
    d_start = torch.from_numpy(start).to(dev)
    # two pose buffers (and gather targets): step k's all-gather runs on RCCL's stream while step k+1 already writes the other buffer
    d_poses2 = [torch.empty((B, ctx.nb, 7), dtype=torch.float32, device=dev) for _ in range(2)]
    d_poses = d_poses2[0]
    d_cnn_out = torch.empty((B, 2304), dtype=torch.float32, device=dev)
    # The exchange: one RCCL all-gather of the poses per step, issued by the LIBRARY (ht_gather_poses_dev: ncclAllGather on the context's communication
    # stream, the way a C++ host shards; the unique id travels through the process group that exists anyway for the barrier).  If the communicator
    # cannot be made, torch.distributed's all-gather does the same exchange and the JSON line says so.
    gather_impl = None
    if use_dist and not cnn_only:
        # every rank goes through the same collectives whatever fails locally, and nobody enters ncclCommInitRank before all ranks can load RCCL (shard.py)
        use_lib, why_not = negotiate_library_gather(dist, dev, rank, world, native.comm_available(), native.comm_unique_id, lambda uid: ctx.comm_init(world, rank, uid))
        if use_lib:
            gather_impl = "ht_gather_poses_dev (ncclAllGather on the context's communication stream): RCCL reports %d rank(s), this is rank %d" % ctx.comm_info()
        else:
            gather_impl = "torch.distributed.all_gather_into_tensor (the library's communicator could not be made on every rank: %s)" % why_not
    use_lib_gather = bool(gather_impl and gather_impl.startswith("ht_"))
    gathered2 = [torch.empty((world * B, ctx.nb, 7), dtype=torch.float32, device=dev) for _ in range(2)] if use_dist else [None, None]
    gathered = gathered2[0]
    stream = torch.cuda.current_stream(dev)
    # the exchange still reading each pose buffer pair: the library's gather is waited for on the stream (ht_gather_wait), torch.distributed's through its work handle
    bufs = PoseBuffers((lambda k, h: ctx.gather_wait(k, stream.cuda_stream)) if use_lib_gather else (lambda k, h: h.wait()))

    w128 = x128 = None
    if e2e:
        ctx.load_weights128(W.make_cnnb128(seed, gain))
    if cnn128:
        w128 = W.make_cnnb128(seed, gain)
        ctx.load_weights128(w128)
        cam_scale = cams[:, 4].astype(np.float32)[:, None, None]
        x128 = np.clip(1.0 - (depth.astype(np.float32) * cam_scale - 0.1) / np.float32(0.7 - 0.1), 0.0, 1.0).astype(np.float32).reshape(B, -1)      # handtrack.h:700 on the whole 128x128 frame
        d_cnn_in = torch.from_numpy(x128).to(dev)
    elif wl == "cnn":
        cnn_in, _, _ = ctx.stage_prepare(depth, cams)
        d_cnn_in = torch.from_numpy(cnn_in).to(dev)

    def step():
        nonlocal d_poses, gathered
        k = bufs.next_slot()       # the exchange of two steps ago has to be through with this buffer pair (it long is)
        d_poses, gathered = d_poses2[k], gathered2[k]
        if cnn128:
            ctx.cnn128_eval_dev(d_cnn_in.data_ptr(), d_cnn_out.data_ptr(), B, stream.cuda_stream)
        elif wl == "cnn":
            ctx.cnn_eval_dev(d_cnn_in.data_ptr(), d_cnn_out.data_ptr(), B, stream.cuda_stream)
        elif e2e:
            ctx.update_direct_dev(d_depth.data_ptr(), d_cams.data_ptr(), 128, d_start.data_ptr(), B, d_poses.data_ptr(), stream.cuda_stream)
        elif cfg5:
            ctx.update_frames_dev(d_depth.data_ptr(), d_cams.data_ptr(), 128, 128, 0.17, d_start.data_ptr(), B, d_poses.data_ptr(), stream.cuda_stream)
        else:
            ctx.update_dev(d_depth.data_ptr(), d_cams.data_ptr(), d_start.data_ptr(), B, d_poses.data_ptr(), stream.cuda_stream)
        if use_lib_gather:
            ctx.gather_poses_dev(d_poses.data_ptr(), gathered.data_ptr(), B, k, stream.cuda_stream)
            bufs.issued(k, True)
        elif use_dist and not cnn_only:
            bufs.issued(k, gather_poses(d_poses, world, out=gathered, force=True, async_op=True)[1])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.profile_enable(1)          # HIP events around the dominant kernel only (on its launch stream), inside the timed region
    ctx.profile_read(reset=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    prof = ctx.profile_read(reset=True)

    # ---- the timed steps' own output against the reference's committed results: every distinct frame of rank 0's batch ----
    verify = None
    if rank == 0 and args.steps > 0:
        if wl == "cnn" and gold is not None:
            got = d_cnn_out[:8].cpu().numpy()
            dc = float(max(np.abs(got[f] - gold["f%d/cnn_output" % f]).max() for f in range(8)))
            verify = {"verified": bool(dc <= VERIFY_CNN_TOL), "against": "tests/golden/golden8.htfx cnn_output (reference), slots 0-7 of the timed batch", "max_abs_dcnn": dc, "tol": VERIFY_CNN_TOL}
        elif cnn128:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as ol
            ref = ol.cnn128_eval(w128, x128[:4])
            dc = float(np.abs(d_cnn_out[:4].cpu().numpy() - ref).max())
            verify = {"verified": bool(dc <= VERIFY_CNN_TOL), "against": "C oracle of the 128x128 net (pinned on the reference's layer classes), frames 0-3 of the timed batch", "max_abs_dcnn": dc, "tol": VERIFY_CNN_TOL}
        elif not cnn_only:
            refp, against, spread = _reference_poses(wl, args.always_take_cnn)
            _pfe, ini = ctx.tracker_flags(B)
            verify = verify_poses(d_poses.cpu().numpy(), ctx.get_state(1, B)[:, :, :7], ini, refp, frame_idx, against, spread, args.always_take_cnn, ill_conditioned=cfg5)
            verify["capacity_events"] = list(ctx.capacity_events())
            if wl == "config5":
                verify["frames_overflow"] = ctx.frames_overflow()
            if use_dist:
                verify["gather_consistent"] = bool(torch.equal(gathered[rank * B:(rank + 1) * B], d_poses))

    # ---- the same steps with the HOST in them (SURVEY 8e names host-side staging as a scaling bound; the application hands update() a host image, synthetic-tracker.cpp:215):
    #      depth and cameras start in pinned host memory, the poses end there; the upload of step k + 1 and the download of step k - 1 run on a copy stream beside step k ----
    host_io = None
    if rank == 0 and wl == "cnn+solver" and args.steps > 0 and not args.no_host_io:
        h_depth = torch.from_numpy(depth.view(np.int16)).pin_memory(); h_cams = torch.from_numpy(cams).pin_memory()
        h_depth2 = [h_depth, torch.from_numpy(depth.view(np.int16).copy()).pin_memory()]      # the zero-copy variant reads the frames where they lie: two pinned buffers, as a capture loop would fill them in turn
        h_poses = [torch.empty((B, ctx.nb, 7), dtype=torch.float32).pin_memory() for _ in range(2)]
        din = [torch.empty_like(d_depth) for _ in range(2)]; cin = [torch.empty_like(d_cams) for _ in range(2)]
        dout = [torch.empty((B, ctx.nb, 7), dtype=torch.float32, device=dev) for _ in range(2)]
        copy = torch.cuda.Stream(device=dev)
        e_in = [torch.cuda.Event() for _ in range(2)]; e_out = [torch.cuda.Event() for _ in range(2)]; e_used = [torch.cuda.Event() for _ in range(2)]

        def upload(k, zero):
            with torch.cuda.stream(copy):
                e_used[k % 2].synchronize()      # the HOST waits until the step that read this buffer pair is through: a host that queues many steps ahead slows the device's own streams down (tools/exp_hostio2.py: 5.63 ms per step queued ahead, 5.10 paced)
                if not zero:
                    din[k % 2].copy_(h_depth, non_blocking=True)
                cin[k % 2].copy_(h_cams, non_blocking=True)      # the cameras (48 B a frame, read by many kernels) are copied in either variant
                e_in[k % 2].record(copy)

        def run(n, transfers, zero=False):
            # zero: no copy engine in the step's way -- a pinned allocation has one address for host and device, so k_prepare reads the frames over the host link itself
            # (ht_update_dev with the pinned pointer) and the step's last kernel writes the poses straight to pinned host memory (tools/exp_hostio_zero_copy.py)
            for b in range(2):
                e_used[b].record(stream)
            if transfers:
                upload(0, zero)
            for k in range(n):
                if transfers and k + 1 < n:
                    upload(k + 1, zero)
                elif not transfers:
                    e_used[(k + 1) % 2].synchronize()      # the same pacing without the transfers: the yardstick of this leg
                if transfers:
                    stream.wait_event(e_in[k % 2])
                ctx.update_dev(h_depth2[k % 2].data_ptr() if zero else din[k % 2].data_ptr(), cin[k % 2].data_ptr(), d_start.data_ptr(), B,
                               h_poses[k % 2].data_ptr() if zero else dout[k % 2].data_ptr(), stream.cuda_stream)
                e_out[k % 2].record(stream); e_used[k % 2].record(stream)
                if transfers and not zero:
                    with torch.cuda.stream(copy):
                        copy.wait_event(e_out[k % 2])
                        h_poses[k % 2].copy_(dout[k % 2], non_blocking=True)
            torch.cuda.synchronize()

        for b in range(2):
            din[b].copy_(d_depth); cin[b].copy_(d_cams)
        ctx.profile_enable(0)
        res = {}
        for transfers in (False, True):
            run(max(2, args.warmup), transfers)
            t0h = time.perf_counter()
            run(args.steps, transfers)
            res[transfers] = time.perf_counter() - t0h
        eh = res[True]
        same = bool(torch.equal(h_poses[(args.steps - 1) % 2], d_poses.cpu()))
        run(max(2, args.warmup), True, True)
        t0h = time.perf_counter()
        run(args.steps, True, True)
        ez = time.perf_counter() - t0h
        same_zero = bool(torch.equal(h_poses[(args.steps - 1) % 2], d_poses.cpu()))
        host_io = {"value": round(B * args.steps / eh, 2), "unit": "frames/s", "ms_per_step": round(eh / args.steps * 1e3, 4),
                   "resident_same_loop_ms_per_step": round(res[False] / args.steps * 1e3, 4), "fraction_of_resident_rate": round(res[False] / eh, 4),
                   "bytes_per_step": {"host_to_device": int(h_depth.numel() * 2 + h_cams.numel() * 4), "device_to_host": int(h_poses[0].numel() * 4)}, "poses_equal_resident_run": same,
                   "what": "pinned host depth + cameras in, poses out to pinned host memory; upload of step k + 1 and download of step k - 1 on a copy stream beside step k (double buffers, the host one step ahead); "
                           "resident_same_loop = the same loop and pacing without the transfers",
                   "zero_copy": {"value": round(B * args.steps / ez, 2), "unit": "frames/s", "ms_per_step": round(ez / args.steps * 1e3, 4), "fraction_of_resident_rate": round(res[False] / ez, 4),
                                 "poses_equal_resident_run": same_zero,
                                 "what": "the same loop without a copy engine: ht_update_dev is handed the PINNED host pointers of the depth frames and of the pose output (one address for host and device); "
                                         "k_prepare reads the frames over the host link, the last solve writes the poses there; only the cameras are copied"}}

    # ---- the same step with TWO batches in flight: a second context on its own streams takes every other batch, so that one batch's kernels fill the gaps the other's
    #      leave (at 1024 frames every kernel of the step is latency-bound and the dominant ones take whole CUs: DESIGN.md section 15).  What a deployment with a queue of
    #      INDEPENDENT batches gets out of the GPU; never `value`: the headline stays one 1024-frame batch at a time ----
    two = None
    if rank == 0 and wl == "cnn+solver" and args.steps > 0 and not args.no_two_in_flight and not use_dist:
        ctx2 = native.Context(os.path.join(ROOT, "hand_tracking_samples_amd", "assets", "model_hand17.htfx"), B, device=local)
        ctx2.load_weights(W.make_cnnb(seed, gain))
        ctx2.set_params(microforce=3.0, mainthreadpasses=3, **({"always_take_cnn": 1} if args.always_take_cnn else {}))
        s2 = torch.cuda.Stream(device=dev)
        out2 = [torch.empty((B, ctx.nb, 7), dtype=torch.float32, device=dev) for _ in range(2)]
        ctx.profile_enable(0)

        def pairs(n):
            for _ in range(n):
                ctx.update_dev(d_depth.data_ptr(), d_cams.data_ptr(), d_start.data_ptr(), B, out2[0].data_ptr(), stream.cuda_stream)
                ctx2.update_dev(d_depth.data_ptr(), d_cams.data_ptr(), d_start.data_ptr(), B, out2[1].data_ptr(), s2.cuda_stream)
            torch.cuda.synchronize()

        pairs(max(2, args.warmup))
        t0p = time.perf_counter()
        pairs(args.steps)
        ep = time.perf_counter() - t0p
        two = {"value": round(2 * B * args.steps / ep, 2), "unit": "frames/s", "ms_per_pair_of_batches": round(ep / args.steps * 1e3, 4), "batches_in_flight": 2, "frames_per_batch": B,
               "poses_equal_single_batch_run": bool(torch.equal(out2[0], d_poses) and torch.equal(out2[1], d_poses)),
               "what": "two contexts, two streams, the same %d-frame batch on each, enqueued alternately: throughput of independent batches when one batch's kernels may run in the gaps of the other's" % B}
        ctx2.close()

    # phase table from a second, untimed pass with every phase bracketed (this serialises the side streams)
    ctx.profile_enable(2)
    nphase = max(2, min(5, args.steps))
    for _ in range(nphase):
        step()
    torch.cuda.synchronize()
    prof_all = ctx.profile_read(reset=True)
    ctx.profile_enable(0)

    if rank == 0:
        total_frames = B * world * args.steps
        value = total_frames / elapsed
        flop_cnn = CNN_FLOP["cnn128" if (cnn128 or e2e) else "cnn"]
        # dominant kernel of the workload by HIP-event time
        phases = {k: v for k, v in prof.items() if v[1] > 0}
        if cnn_only:
            phases = {k: (v[0] * args.steps / nphase, v[1] * args.steps // nphase) for k, v in prof_all.items() if v[1] > 0}
        all_phases = {k: v for k, v in prof_all.items() if v[1] > 0}
        dom = max(all_phases, key=lambda k: all_phases[k][0] / nphase) if all_phases else None
        if dom not in phases:
            phases[dom] = (all_phases[dom][0] * args.steps / nphase, all_phases[dom][1] * args.steps // nphase)
        roof = None
        if dom == "solve":
            # One k_solve launch per frame.  HBM side: the body state in and out (2 x nb x 52 B) plus the constraint rows it
            # consumes (64 B each).  VALU side: 20 sweeps x (120 flop per linear row + 60 per angular row).  Row counts per frame from the data.
            scale = cams[:, 4].astype(np.float32)[:, None]
            z = depth.reshape(B, -1).astype(np.float32) * scale
            npts = (((z >= 0.1) & (z < 0.7)).sum(axis=1) + 3) // 4
            main_rows = float(np.mean(npts + np.where(npts > 400, 5 * ctx.nb, 0))); sim_rows = float(np.mean((npts + 3) // 4))
            rows_mean = (3 * main_rows + 4 * sim_rows) / 8.0              # 3 main passes, 4 cloud-bearing MultiStepSim steps (+ 1 without cloud rows)
            joint_rows, ang_rows, contact_rows = 3.0 * ctx.nj, 71.0 * ctx.nj / 16.0, 3 * 5.1      # 71 angular rows, 5.1 contacts per frame (17-bone hand)
            per_frame = 2 * ctx.nb * 52 + 64.0 * rows_mean
            flop_frame = 20.0 * ((rows_mean + joint_rows + contact_rows) * 120.0 + ang_rows * 60.0)
            avg_ms = phases[dom][0] / phases[dom][1]
            achieved = per_frame * B / (avg_ms * 1e-3) / 1e9
            valu = flop_frame * B / (avg_ms * 1e-3) / 1e12
            # HBM bytes per launch from the committed rocprofv3 PMC passes of THIS workload and batch size (profiles/rNN_pmc_hbm_traffic*.json carry
            # what they were measured on); null when no such measurement exists
            traffic = _pmc_traffic("k_solve", wl, B)
            roof = {"kernel": "k_solve", "bound": "latency", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                    "hbm_frac": round(achieved / HBM_PEAK_GBS, 6), "valu_achieved_tflops": round(valu, 3), "valu_peak_tflops": FP32_VALU_PEAK_TF, "valu_frac": round(valu / FP32_VALU_PEAK_TF, 6),
                    "traffic": traffic, "avg_launch_ms": round(avg_ms, 4), "launches": phases[dom][1], "algorithmic_bytes_per_launch": int(per_frame * B), "algorithmic_flop_per_launch": int(flop_frame * B),
                    "note": "sequential Gauss-Seidel (physics.h:556-562): bound by the dependent-instruction latency of the longest per-body row chain, neither by HBM nor by VALU throughput; both fractions are reported (SURVEY 8d); "
                            "issue_frac = VALU instructions issued x 4 clocks / the resident waves' clocks (one wave per SIMD at 1024 frames), from the committed counter pass"}
            roof.update(_pmc_issue("k_solve", wl, B))
        elif dom == "contacts":
            # k_contacts reads the poses of the frame's bodies and writes its contacts (48 B each, data dependent, not counted): like the
            # solve it is a latency-bound kernel (one wave walks the candidate pairs' GJK/EPA iterations), priced against HBM for the record
            per_frame = ctx.nb * 28 + 4
            avg_ms = phases[dom][0] / phases[dom][1]
            achieved = per_frame * B / (avg_ms * 1e-3) / 1e9
            roof = {"kernel": "k_contacts", "bound": "latency", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                    "avg_launch_ms": round(avg_ms, 4), "launches": phases[dom][1], "algorithmic_bytes_per_launch": int(per_frame * B),
                    "note": "broad phase + GJK/EPA over %d body pairs per frame: latency/LDS-bound, not a streaming kernel" % (ctx.nb * (ctx.nb - 1) // 2)}
        elif dom in ("cnn", "cnn128"):
            avg_ms = phases[dom][0] / phases[dom][1]
            achieved = flop_cnn * B / (avg_ms * 1e-3) / 1e12
            roof = {"kernel": "cnn (k_conv12 + k_fc + k_fc144_pk + k_softmax_decode)" if dom == "cnn" else "cnn (k_conv1 + k_conv2 + k_fc + k_fc144_pk + k_softmax_decode)", "bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK_TF, 5), "traffic": None, "avg_launch_ms": round(avg_ms, 4), "launches": phases[dom][1], "flop_per_frame": flop_cnn}
        cnn_roof = None
        cnn_phase = "cnn128" if "cnn128" in all_phases else "cnn"
        if cnn_phase in all_phases and dom != cnn_phase:
            avg_ms = all_phases[cnn_phase][0] / all_phases[cnn_phase][1]
            ach = CNN_FLOP[cnn_phase] * B / (avg_ms * 1e-3) / 1e12
            cnn_roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TF, 5), "avg_ms": round(avg_ms, 4)}
        shard_note = ("; layout of BASELINE configs[3] (contiguous shard per GPU, RCCL all-gather of the poses inside the timed step; configs[3] itself = --gpus 8 --frames-per-gpu 8192)"
                      if world > 1 else "")
        workloads = {
            "cnn+solver": ("synthetic depth frames/sec (CNN+solver), 64x64x1 input, 17-bone hand",
                           "BASELINE configs[2]: %d independent 64x64 frames per GPU, CNN + decode + 5-step MultiStepSim + 3 FitPointCloud passes (GJK + PGS), 17 bones%s" % (B, shard_note)),
            "cnn": ("synthetic depth frames/sec (CNN forward only), 64x64x1 input", "BASELINE configs[1]: %d frames per GPU, CNN forward only" % B),
            "config5": ("synthetic depth frames/sec (segmentation+CNN+solver), 128x128x1 input, 26-bone hand",
                        "BASELINE configs[4]: %d independent 128x128 frames per GPU, HandTracker::update on full frames (HandSegmentVR + CNN + 5-step MultiStepSim + 3 FitPointCloud passes), 26 bones%s" % (B, shard_note)),
            "config5-e2e": ("synthetic depth frames/sec (CNN+solver), 128x128x1 input, 26-bone hand",
                            "BASELINE configs[4] end to end (SURVEY 8d config 5 i-iii): %d independent 128x128 frames per GPU, the 128x128-input net (conv5x5 @124, conv4x4 @28, FC 12544->2048->2304) -> CNNOutputAnalysis(camsub 8) -> "
                            "FitError / reset / 5-step MultiStepSim / accept -> 3 FitPointCloud passes (GJK + PGS), 26 bones, no segmentation%s" % (B, shard_note)),
            "config5-cnn128": ("synthetic depth frames/sec (CNN forward only), 128x128x1 input",
                               "BASELINE configs[4], CNN part at full input size (SURVEY 8d config 5 ii): %d frames per GPU, conv5x5 1->16 @124, 2x pool, conv4x4 16->64 @28, pool, FC 12544->2048->2304, chunked softmax" % B),
        }
        out = {
            "metric": workloads[wl][0],
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": ("synthetic (%d distinct software-rendered animbank frames%s; seeded weights 0x5EED0001)"
                                      % (min(ndistinct, B * world), "" if B * world <= ndistinct else ", each %d times in the global batch" % ((B * world + ndistinct - 1) // ndistinct))),
            "config": {"workload": workloads[wl][1], "frames_per_gpu": B, "global_frames_per_step": B * world,
                       "parallelism": ("frames sharded per GPU, RCCL all-gather of poses" if not cnn_only else "frames sharded per GPU, no exchange") if world > 1 else "single GPU"},
            **({"gather": gather_impl} if gather_impl else {}),
            "roofline": roof,
            "phase_ms_per_step": {k: round(v[0] / nphase, 4) for k, v in sorted(all_phases.items())},
            "phase_note": "from an extra untimed pass with every phase bracketed and the side streams serialised",
        }
        if bad:
            out["tuning_run_not_a_result"] = bad
        if verify is not None:
            out["verified"] = verify["verified"]
            out["verify"] = verify
        if cnn_roof:
            out["roofline_cnn"] = cnn_roof
        if host_io:
            out["host_io"] = host_io
        if two:
            out["two_batches_in_flight"] = two
        if world == 1 and not args.no_cpu_baseline:
            if cnn128:
                out["cpu_baseline"] = cpu_baseline_cnn128(x128, w128)
            else:
                out["cpu_baseline"] = (cpu_baseline_e2e128 if e2e else cpu_baseline_config5 if cfg5 else cpu_baseline)(depth, cams, start, seed, gain)
                if wl == "cnn" and "cnn_only_fps" in out["cpu_baseline"]:
                    out["cpu_baseline"]["frame_fps_cnn_plus_solver"] = out["cpu_baseline"]["value"]
                    out["cpu_baseline"]["value"] = out["cpu_baseline"].pop("cnn_only_fps")
                    out["cpu_baseline"]["sample"] += "; value = CNN::Eval only"
            out["speedup_vs_cpu_1thread"] = round(value / out["cpu_baseline"]["value"], 1)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if verify is not None and verify.get("verified") is False and not bad:
        sys.stderr.write("bench.py: the timed steps' output does NOT match the reference's committed results: %s\n" % json.dumps(verify))
        sys.exit(3)


if __name__ == "__main__":
    main()
