"""Init-time model build: ht_model_bake / ht_create(model.json) against what the reference's own
PhysModel constructor (+ LoadHandModel) produced.  Every array must match bit for bit.

  * model_chain3.json  -- our own 3-body file (tests/golden/make_model_chain3.py); expected build from
                          `oracle/_ref/ref_harness modelfile` (PhysModel(const char*), physmodel.h:444-475).
  * model_hand.json    -- the reference asset, only readable in the build container; expected build is the committed
                          model_hand17.htfx (`ref_harness model`, i.e. after LoadHandModel, handtrack.h:347-366).
"""
import os

import numpy as np
import pytest

import htfx
from hand_tracking_samples_amd import native

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF_HAND_JSON = "/root/reference/assets/model_hand.json"


def assert_same_model(built, expected):
    assert set(expected) <= set(built), sorted(set(expected) - set(built))
    for k, e in expected.items():
        b = built[k]
        assert b.dtype == e.dtype and b.shape == e.shape, (k, b.dtype, b.shape, e.dtype, e.shape)
        assert b.tobytes() == e.tobytes(), "%s differs (max abs %.3g)" % (k, np.abs(b.astype(np.float64) - e.astype(np.float64)).max())


def test_chain3_matches_reference_build(tmp_path):
    out = tmp_path / "chain3.htfx"
    native.model_bake(os.path.join(GOLD, "model_chain3.json"), out, hand_tweaks=False)
    built, expected = htfx.load(str(out)), htfx.load(os.path.join(GOLD, "model_chain3.htfx"))
    assert_same_model(built, expected)
    assert list(built["nverts"]) == [98, 74, 322] and list(built["nplanes"]) == [92, 92, 92]      # 48-vertex hulls


def test_chain3_hand_tweaks_only_touch_collision_vertices(tmp_path):
    a, b = tmp_path / "a.htfx", tmp_path / "b.htfx"
    native.model_bake(os.path.join(GOLD, "model_chain3.json"), a, hand_tweaks=False)
    native.model_bake(os.path.join(GOLD, "model_chain3.json"), b, hand_tweaks=True)
    A, B = htfx.load(str(a)), htfx.load(str(b))
    for k in A:
        if k == "b2/verts":      # handtrack.h:350-352: bodies >= 2 are shrunk by (0.7, 0.7, 0.9) after planes/radii were taken
            assert np.array_equal(B[k], A[k] * np.array([0.7, 0.7, 0.9], np.float32))
        else:
            assert A[k].tobytes() == B[k].tobytes(), k


@pytest.mark.skipif(not os.path.exists(REF_HAND_JSON), reason="reference asset only exists in the build container")
def test_hand_model_matches_reference_build(tmp_path):
    out = tmp_path / "hand.htfx"
    native.model_bake(REF_HAND_JSON, out, hand_tweaks=True)
    assert_same_model(htfx.load(str(out)), htfx.load(os.path.join(GOLD, "model_hand17.htfx")))


@pytest.mark.skipif(not os.path.exists(REF_HAND_JSON), reason="reference asset only exists in the build container")
def test_hand26_model_matches_reference_build(tmp_path):
    """BASELINE configs[4] model (tests/golden/make_model_hand26.py): 26 bones, built by our builder and by the reference's constructor."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_model_hand26", os.path.join(GOLD, "make_model_hand26.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    js, out = tmp_path / "hand26.json", tmp_path / "hand26.htfx"
    mod.main(REF_HAND_JSON, str(js))
    native.model_bake(str(js), out, hand_tweaks=True)
    built, expected = htfx.load(str(out)), htfx.load(os.path.join(GOLD, "model_hand26.htfx"))
    assert_same_model(built, expected)
    assert len(built["nverts"]) == 26


@pytest.mark.parametrize("text", [
    "",                                                                  # empty file
    "[1, 2, 3]",                                                         # not an object
    '{"controlcages": [], "joints": []}',                                # no bodies
    '{"controlcages": [{"verts": [[0,0,0],[1,0,0],[0,1,0]], "faces": [[0,1,2]]}], "joints": []}',      # open mesh
    '{"controlcages": [{"verts": [[0,0,0]], "faces": [[0,1,2]]}], "joints": []}',                      # index out of range
    '{"controlcages": [{"verts": [], "faces": []}, {"verts": [], "faces": []}], "joints": [{"rbi0": 1, "rbi1": 0}]}',
])
def test_malformed_models_are_rejected(tmp_path, text):
    src = tmp_path / "bad.json"
    src.write_text(text)
    with pytest.raises(native.HTError):
        native.model_bake(src, tmp_path / "bad.htfx")
    assert not (tmp_path / "bad.htfx").exists()


def test_missing_file_is_rejected(tmp_path):
    with pytest.raises(native.HTError):
        native.model_bake(tmp_path / "nope.json", tmp_path / "x.htfx")


@pytest.mark.gpu
def test_create_from_json_builds_on_the_fly(tmp_path):
    ctx = native.Context(os.path.join(GOLD, "model_chain3.json"), max_batch=1)
    try:
        assert (ctx.nb, ctx.nj) == (3, 2)
    finally:
        ctx.close()


def test_shipped_model_assets_are_what_the_reference_builds():
    """bench.py and smoke() run on hand_tracking_samples_amd/assets/model_hand*.htfx, baked by the product's own builder (tools/bake_assets.sh): every array
    the reference's constructor produced (the fixtures under tests/golden/) is in them, bit for bit."""
    import htfx
    for n in ("17", "26"):
        ours = htfx.load(os.path.join(os.path.dirname(HERE), "hand_tracking_samples_amd", "assets", "model_hand%s.htfx" % n))
        ref = htfx.load(os.path.join(HERE, "golden", "model_hand%s.htfx" % n))
        shared = [k for k in ref if k in ours]
        assert len(shared) >= 60, "only %d arrays in common" % len(shared)
        for k in shared:
            assert np.array_equal(ours[k], ref[k]), "model_hand%s: %s" % (n, k)
