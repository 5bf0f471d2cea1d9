"""How far does the REFERENCE move between its own builds?  (build container only: needs /root/reference)

Compiles oracle/ref_harness.cpp against the reference headers four ways -- IEEE (-O2 -ffp-contract=off: the build every fixture comes from),
FMA-contracted (-O2 -march=native, clang's default -ffp-contract=on, and =fast) and the reference Makefile's own default (-Ofast -march=native,
Makefile:22-28) -- runs the whole unit of work on the 1024 bench frames (tests/golden/frames1024.npz) with each and prints / writes the per-frame
pose differences against the IEEE build.  This is the yardstick for the floating-point tolerance of the device path: the device solver evaluates
the reference's row updates in Jacobian form with fused multiply-adds (csrc/ht_quad.hpp), i.e. it is one more "build" of the same algorithm.

    python tests/golden/ref_flag_spread.py [out.json [per_frame.npz [takecnn]]]

With `takecnn` the unit of work runs with always_take_cnn = 1 (every CNN-driven pose accepted: the user pose then depends on the net and MultiStepSim on every frame);
the per-frame file is committed as tests/golden/ref_spread1024_takecnn.npz and held against tests/golden/poses1024_takecnn.htfx.
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
    take = len(sys.argv) > 3 and sys.argv[3] == "takecnn"
    with tempfile.TemporaryDirectory() as td:
        d = np.load(os.path.join(HERE, "frames1024.npz"))
        frames = os.path.join(td, "frames1024.htfx")
        htfx.save(frames, {"depth": d["depth"].reshape(-1, 64, 64), "cam": d["cam"], "startpose": d["startpose"]})
        res = {}
        for name, flags in BUILDS.items():
            exe = os.path.join(td, "ref_%d" % len(res))
            subprocess.check_call([CLANG, "-std=c++14"] + flags + ["-Wno-narrowing", "-fdelayed-template-parsing", "-w", "-I" + os.path.join(ROOT, "oracle"),
                                   os.path.join(ROOT, "oracle", "ref_harness.cpp"), "-o", exe, "-lpthread"])
            subprocess.check_call([exe, "poses", frames, "0x5EED0001", "24", exe + ".htfx"] + (["takecnn"] if take else []))
            res[name] = htfx.load(exe + ".htfx")
        committed = htfx.load(os.path.join(HERE, "poses1024_takecnn.htfx" if take else "poses1024.htfx"))
        assert np.array_equal(committed["uw_pose_user"], res["ieee"]["uw_pose_user"]), "the committed poses fixture is not what the IEEE build produces"
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
        json.dump({"what": "reference built with other compiler flags vs its IEEE build (-O2 -ffp-contract=off), whole unit of work on the 1024 bench frames (tests/golden/frames1024.npz)", "builds": out}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
