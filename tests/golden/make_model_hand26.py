"""BASELINE configs[4] model: the 17-bone hand plus three more 3-bone fingers = 26 bones ("config 5").

The reference ships no such asset, so this is a build-side definition: the index, middle and ring chains (bones 5-7, 8-10, 11-13 of
assets/model_hand.json) are cloned as bones 17-19, 20-22, 23-25 and attached to the palm DY = 3 cm to the dorsal side of the originals
(palm-local +y), same cages, same joint frames and ranges.  Bones 0-16 keep their indices, so everything in handtrack.h that names
bones by number (landmarks :77-81, LoadHandModel :347-366, HandModelEnhancements :406-441) keeps its meaning.  The file is written in
the reference's own JSON schema so that PhysModel(const char*) (physmodel.h:444) loads it inside oracle/_ref/ref_harness.

    python tests/golden/make_model_hand26.py /tmp/model_hand26.json          # needs /root/reference/assets/model_hand.json
    HT_REF_MODEL_JSON=/tmp/model_hand26.json oracle/_ref/ref_harness model tests/golden/model_hand26.htfx

Only the built arrays (model_hand26.htfx, as the reference's constructor + LoadHandModel produce them) are committed, not the JSON.
A pose of the 26-bone hand follows from a 17-bone pose: clone bone = source bone shifted by R(palm) * (0, DY, 0), same orientation.
"""
import json
import sys

import numpy as np

DY = 0.03
CLONES = [5, 6, 7, 8, 9, 10, 11, 12, 13]      # source bone of bones 17..25


def qrot(q, v):
    x, y, z, w = q
    u = np.array([x, y, z], np.float64)
    return v + 2.0 * np.cross(u, np.cross(u, v) + w * v)


def extend_pose(pose17):
    """[17][7] -> [26][7] (float64 arithmetic, rounded by the caller)"""
    p = np.asarray(pose17, np.float64)
    off = qrot(p[1, 3:7], np.array([0.0, DY, 0.0]))
    ext = [np.concatenate([p[s, 0:3] + off, p[s, 3:7]]) for s in CLONES]
    return np.concatenate([p, np.array(ext)], 0)


def main(src, dst):
    m = json.load(open(src))
    remap = {s: 17 + i for i, s in enumerate(CLONES)}
    for s in CLONES:
        m["controlcages"].append(json.loads(json.dumps(m["controlcages"][s])))
    for j in list(m["joints"]):
        if j["rbi1"] in remap:
            k = json.loads(json.dumps(j))
            k["rbi1"] = remap[j["rbi1"]]
            if j["rbi0"] == 1:
                k["p0"] = [float("%.6g" % (j["p0"][0])), float("%.6g" % (j["p0"][1] + DY)), float("%.6g" % (j["p0"][2]))]
            else:
                k["rbi0"] = remap[j["rbi0"]]
            m["joints"].append(k)
    m["pose"] = [[float("%.6g" % x) for x in row] for row in extend_pose(m["pose"])]
    with open(dst, "w") as fp:
        json.dump(m, fp, indent=1)
    print("%s: %d cages, %d joints" % (dst, len(m["controlcages"]), len(m["joints"])))


if __name__ == "__main__":
    main("/root/reference/assets/model_hand.json", sys.argv[1] if len(sys.argv) > 1 else "/tmp/model_hand26.json")
