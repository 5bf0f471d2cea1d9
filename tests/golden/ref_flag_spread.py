"""How far does the REFERENCE move between its own builds?  (build container only: needs /root/reference)

Compiles oracle/ref_harness.cpp against the reference headers four ways -- IEEE (-O2 -ffp-contract=off: the build every fixture comes from),
FMA-contracted (-O2 -march=native, clang's default -ffp-contract=on, and =fast) and the reference Makefile's own default (-Ofast -march=native,
Makefile:22-28) -- runs the whole unit of work on the 1024 bench frames (bench_data/frames1024.npz) with each and prints / writes the per-frame
pose differences against the IEEE build.  This is the yardstick for the floating-point tolerance of the device path: the device solver evaluates
the reference's row updates in Jacobian form with fused multiply-adds (csrc/ht_quad.hpp), i.e. it is one more "build" of the same algorithm.

    python tests/golden/ref_flag_spread.py [out.json [per_frame.npz [takecnn | config5 | e2e]]]

With `takecnn` the unit of work runs with always_take_cnn = 1 (every CNN-driven pose accepted: the user pose then depends on the net and MultiStepSim on every frame);
the per-frame file is committed as tests/golden/ref_spread1024_takecnn.npz and held against tests/golden/poses1024_takecnn.htfx.
`config5` / `e2e`: BASELINE configs[4] instead -- the 256 frames of bench_data/frames5_256.npz with the 26-bone hand as the reference runs them (posesfull) / end to end
with the 128x128-input net (e2e128); committed as tests/golden/ref_spread5_256.npz / ref_spread5e2e_256.npz.
The FMA builds are compiled for -march=x86-64-v3 (a fixed target: the yardstick does not depend on the host that regenerates it).

per_frame.npz (committed as tests/golden/ref_spread1024.npz): for every one of the 1024 bench frames how far the reference's two FMA-contracted builds move
from its IEEE build (|dpos|, |dquat| of the user poses and of othermodel) -- a frame's SENSITIVITY to rounding, which tests/test_gpu_batch_parity.py holds the
device's own deviation against.

Test infrastructure only; nothing here is used by the product path.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import htfx  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
BUILDS = {"ieee": ["-O2", "-ffp-contract=off"], "fma_on": ["-O2", "-march=x86-64-v3", "-ffp-contract=on"], "fma_fast": ["-O2", "-march=x86-64-v3", "-ffp-contract=fast"],
          "Ofast (reference Makefile default)": ["-Ofast", "-march=x86-64-v3"]}


def spread(a, b):
    dp = np.abs(a[:, :, :3] - b[:, :, :3]).max(axis=(1, 2))
    dq = np.minimum(np.abs(a[:, :, 3:7] - b[:, :, 3:7]), np.abs(a[:, :, 3:7] + b[:, :, 3:7])).max(axis=(1, 2))
    pct = lambda v: [float(x) for x in np.percentile(v, [50, 90, 99, 100])]
    return {"frames": int(len(dp)), "exact": int(((dp == 0) & (dq == 0)).sum()), "within_2e-5m_2e-4": int(((dp <= 2e-5) & (dq <= 2e-4)).sum()),
            "within_2e-4m_2e-3": int(((dp <= 2e-4) & (dq <= 2e-3)).sum()), "dpos_p50_p90_p99_max": pct(dp), "dquat_p50_p90_p99_max": pct(dq),
            "worst_frames": [int(i) for i in np.argsort(dp)[-3:]]}


def main():
    if not os.path.isdir("/root/reference/include"):
        sys.exit("reference tree not present: this measurement runs in the build container only")
    out = {}
    mode = sys.argv[3] if len(sys.argv) > 3 else ""
    take = mode == "takecnn"
    five = mode in ("config5", "e2e")      # BASELINE configs[4]: the 256 frames of 128x128 with the 26-bone hand, as the reference runs them (posesfull) / end to end (e2e128)
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ)
        if five:
            d = np.load(os.path.join(ROOT, "bench_data", "frames5_256.npz"))
            frames = os.path.join(td, "frames5_256.htfx")
            htfx.save(frames, {"depth": d["depth"], "cam": d["cam"], "startpose": d["startpose"], "rows": d["rows"]})
            mj = os.path.join(td, "model_hand26.json")
            subprocess.check_call([sys.executable, os.path.join(HERE, "make_model_hand26.py"), mj], stdout=subprocess.DEVNULL)
            env["HT_REF_MODEL_JSON"] = mj
        else:
            d = np.load(os.path.join(ROOT, "bench_data", "frames1024.npz"))
            frames = os.path.join(td, "frames1024.htfx")
            htfx.save(frames, {"depth": d["depth"].reshape(-1, 64, 64), "cam": d["cam"], "startpose": d["startpose"]})
        res = {}
        for name, flags in BUILDS.items():
            exe = os.path.join(td, "ref_%d" % len(res))
            subprocess.check_call([CLANG, "-std=c++14"] + flags + ["-Wno-narrowing", "-fdelayed-template-parsing", "-w", "-I" + os.path.join(ROOT, "oracle"),
                                   os.path.join(ROOT, "oracle", "ref_harness.cpp"), "-o", exe, "-lpthread"])
            if mode == "config5":
                subprocess.check_call([exe, "posesfull", frames, "0x5EED0001", "24", exe + ".htfx"], env=env, stdout=subprocess.DEVNULL)
                res[name] = htfx.load(exe + ".htfx")
            elif mode == "e2e":
                subprocess.check_call([exe, "e2e128", frames, "", "0x5EED0001", "24", exe + ".htfx"], env=env, stdout=subprocess.DEVNULL)
                r = htfx.load(exe + ".htfx"); res[name] = {"uw_pose_user": r["all/uw_pose_user"], "other_pose": r["all/other_pose"]}
            else:
                subprocess.check_call([exe, "poses", frames, "0x5EED0001", "24", exe + ".htfx"] + (["takecnn"] if take else []))
                res[name] = htfx.load(exe + ".htfx")
        if mode == "config5":
            committed = htfx.load(os.path.join(HERE, "poses5full256.htfx"))["uw_pose_user"]
        elif mode == "e2e":
            committed = htfx.load(os.path.join(HERE, "e2e128_256.htfx"))["all/uw_pose_user"]
        else:
            committed = htfx.load(os.path.join(HERE, "poses1024_takecnn.htfx" if take else "poses1024.htfx"))["uw_pose_user"]
        assert np.array_equal(committed, res["ieee"]["uw_pose_user"]), "the committed poses fixture is not what the IEEE build produces"
        per_frame = {}
        for name in list(BUILDS)[1:]:
            out[name] = {"handmodel_user_pose": spread(res["ieee"]["uw_pose_user"], res[name]["uw_pose_user"]), "othermodel_pose": spread(res["ieee"]["other_pose"], res[name]["other_pose"])}
            print(name, json.dumps(out[name]))
            if name.startswith("fma"):
                for key, arr in (("user", "uw_pose_user"), ("other", "other_pose")):
                    a, b = res["ieee"][arr], res[name][arr]
                    per_frame["%s_%s_dpos" % (name, key)] = np.abs(a[:, :, :3] - b[:, :, :3]).max(axis=(1, 2)).astype(np.float32)
                    per_frame["%s_%s_dquat" % (name, key)] = np.minimum(np.abs(a[:, :, 3:7] - b[:, :, 3:7]), np.abs(a[:, :, 3:7] + b[:, :, 3:7])).max(axis=(1, 2)).astype(np.float32)
        if len(sys.argv) > 2:
            np.savez_compressed(sys.argv[2], **per_frame)
    if len(sys.argv) > 1:
        json.dump({"what": "reference built with other compiler flags vs its IEEE build (-O2 -ffp-contract=off), whole unit of work on the bench frames (%s)" % ("bench_data/frames5_256.npz, 26 bones, " + mode if five else "bench_data/frames1024.npz"), "builds": out}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
