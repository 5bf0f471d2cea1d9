"""GPU parity of the CNN path (k_prepare, k_conv1, k_conv2, k_fc, k_softmax_decode) against the CPU oracle and the
committed golden vectors, through the C-ABI.  Needs an MI355X: pytest -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_lib as ol

# CNN output tolerance: fp32 MFMA accumulates with one rounding per multiply-add (fused) where the reference rounds the product
# and the sum separately (cnn.h:242,420); over K<=2304 terms that is a few 1e-6 relative on the logits.  Outputs are softmax
# probabilities <= 1, so an absolute bound is used.
CNN_ATOL = 2e-5


@pytest.fixture(scope="module")
def ctx(weights):
    from hand_tracking_samples_amd import native
    c = native.Context(ol.MODEL, 64)
    c.load_weights(weights)
    yield c
    c.close()


def _frames(golden, n=8):
    depth = np.stack([golden["f%d/depth" % f].reshape(-1) for f in range(n)])
    cams = np.stack([golden["f%d/cam" % f] for f in range(n)])
    return depth, cams


def test_prepare_matches_golden(ctx, golden):
    depth, cams = _frames(golden)
    cnn_in, pts, n = ctx.stage_prepare(depth, cams)
    for f in range(8):
        pre = "f%d/" % f
        assert n[f] == golden[pre + "pc_count"][1]
        assert np.array_equal(pts[f, :n[f], :3], golden[pre + "vpts"])          # bit exact: same IEEE ops as misc_image.h:48,409-417
        if (pre + "cnn_input") in golden:
            assert np.array_equal(cnn_in[f], golden[pre + "cnn_input"])


def test_cnn_eval_matches_golden(ctx, golden):
    depth, cams = _frames(golden)
    cnn_in, _, _ = ctx.stage_prepare(depth, cams)
    out = ctx.cnn_eval(cnn_in)
    ref = np.stack([golden["f%d/cnn_output" % f] for f in range(8)])
    err = np.abs(out - ref).max()
    print("cnn max abs err vs reference golden: %.3e" % err)
    assert err <= CNN_ATOL
    assert np.allclose(out.sum(axis=1), 24.0, atol=1e-3)


def test_cnn_layers_match_the_references_layer_outputs(ctx, golden, weights):
    """Each kernel of the forward pass against the layer outputs of the reference (cnn.h:550-556 returns every layer's vector): k_conv1 = layers 0-3
    (conv 5x5, tanh, two pools), k_conv2 = layers 4-6, k_fc<tanh> = layers 7-8, k_fc144 = layer 9, soft-max = layer 10.  Frame 0 against the
    reference's own dump in the fixture, all eight frames against the C restatement (which reproduces that dump bit for bit,
    test_oracle_vs_golden.py::test_cnn_layers_bit_exact).  A compensating error in two layers cannot pass here."""
    depth, cams = _frames(golden)
    cnn_in, _, _ = ctx.stage_prepare(depth, cams)
    out = ctx.cnn_eval(cnn_in)
    a1, a2, a3, lg = ctx.cnn_layers(8)
    names = {3: "conv1+tanh+pool+pool", 6: "conv2+tanh+pool", 8: "fc1+tanh", 9: "fc2", 10: "softmax"}
    sizes = (57600, 57600, 14400, 3600, 9216, 9216, 2304, 2048, 2048, 2304, 2304)
    L = ol.lib()
    worst = {}
    for f in range(8):
        layers = [np.zeros(n, np.float32) for n in sizes]
        arr = (C.POINTER(C.c_float) * 11)(*[ol.fptr(a) for a in layers])
        ref_out = np.zeros(2304, np.float32)
        L.ho_cnn_eval(ol.fptr(weights), ol.fptr(np.ascontiguousarray(cnn_in[f])), ol.fptr(ref_out), arr)
        got = {3: a1[f], 6: a2[f], 8: a3[f], 9: lg[f], 10: out[f]}
        for li, name in names.items():
            if f == 0:
                assert np.array_equal(layers[li], golden["f0/cnn_layer%d" % li])      # the checker is the reference's dump on this frame
            worst[name] = max(worst.get(name, 0.0), float(np.abs(got[li] - layers[li]).max()))
    print("per-layer max abs err vs the reference:", {k: "%.2e" % v for k, v in worst.items()})
    # activations are tanh outputs in [-1, 1] (absolute bound as for the output); the logits reach a few units times the gain of 24
    assert worst["conv1+tanh+pool+pool"] <= 2e-6 and worst["conv2+tanh+pool"] <= 5e-6 and worst["fc1+tanh"] <= 1e-5 and worst["fc2"] <= 1e-4 and worst["softmax"] <= CNN_ATOL


def test_cnn_eval_full_batch_vs_oracle(weights):
    """ht_cnn_eval at the batch the last fully connected layer is tiled for (1024 frames = 16 x 16 blocks of 64 x 144): a strided sample of 64 frames
    against the C restatement, and every copy of a frame bit-identical to the others (a frame's result does not depend on its slot)."""
    from hand_tracking_samples_amd import native
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames256.npz"))
    B = 1024
    idx = np.arange(B) % 256
    c = native.Context(ol.MODEL, B)
    try:
        c.load_weights(weights)
        cnn_in, _, _ = c.stage_prepare(d["depth"][idx].reshape(B, -1), d["cam"][idx])
        out = c.cnn_eval(cnn_in)
    finally:
        c.close()
    for k in range(1, 4):
        assert np.array_equal(out[:256], out[256 * k:256 * (k + 1)])
    L = ol.lib()
    worst = 0.0
    for i in range(0, B, 16):
        ref = np.zeros(2304, np.float32)
        L.ho_cnn_eval(ol.fptr(weights), ol.fptr(np.ascontiguousarray(cnn_in[i])), ol.fptr(ref), None)
        worst = max(worst, float(np.abs(out[i] - ref).max()))
    print("cnn at B=1024, 64 sampled frames: max abs err %.3e" % worst)
    assert worst <= CNN_ATOL


def test_cnn_eval_random_inputs_vs_oracle(ctx, weights):
    """Seeded random tiles (including all-zero and all-one inputs) at a batch that is not a multiple of the GEMM tile."""
    rng = np.random.default_rng(7)
    B = 37
    x = rng.random((B, 4096), dtype=np.float32)
    x[0] = 0.0
    x[1] = 1.0
    out = ctx.cnn_eval(x)
    L = ol.lib()
    ref = np.zeros((B, 2304), np.float32)
    for i in range(B):
        L.ho_cnn_eval(ol.fptr(weights), ol.fptr(np.ascontiguousarray(x[i])), ol.fptr(ref[i]), None)
    err = np.abs(out - ref).max()
    print("cnn max abs err vs oracle on random inputs: %.3e" % err)
    assert err <= CNN_ATOL


def test_the_net_inside_an_update_equals_the_net_alone(ctx, golden):
    """ht_launch_cnn picks its launch arrangement by context: an update, whose side branch runs beside the net, keeps the two convolution launches (k_conv1, k_conv2); a
    stand-alone evaluation takes the fused one (k_conv12).  Same arithmetic in the same order: the heat-maps an update returns equal ht_cnn_eval's on the same tiles bit
    for bit, and so do the layers in between."""
    depth, cams = _frames(golden)
    start = np.stack([golden["f%d/startpose" % f] for f in range(8)]) if "f0/startpose" in golden else None
    if start is None:
        start = np.zeros((8, 17, 7), np.float32); start[:, :, 6] = 1.0; start[:, :, 2] = 0.45
    ctx.tracker_reset(start)
    _, cnn_update = ctx.update_sync(depth, cams, want_cnn=True)
    layers_update = ctx.cnn_layers(8)
    cnn_in, _, _ = ctx.stage_prepare(depth, cams)
    cnn_alone = ctx.cnn_eval(cnn_in)
    layers_alone = ctx.cnn_layers(8)
    assert np.array_equal(cnn_update, cnn_alone)
    for a, b in zip(layers_update, layers_alone):
        assert np.array_equal(a, b)


def test_decode_matches_golden(ctx, golden):
    depth, cams = _frames(golden)
    cnn_out = np.stack([golden["f%d/cnn_output" % f] for f in range(8)])
    an = ctx.stage_decode(cnn_out, cams)
    for f in range(8):
        pre = "f%d/" % f
        # teacher-forced on the reference's own CNN output: every stage but sin/cos is the same IEEE arithmetic
        assert np.array_equal(an[f, 0:32].reshape(8, 4), golden[pre + "an_crays"])
        assert np.array_equal(an[f, 32:48].reshape(8, 2), golden[pre + "an_image_points"])
        assert np.array_equal(an[f, 48:56], golden[pre + "an_confidence"])
        assert np.array_equal(an[f, 56:72], golden[pre + "an_vals"])
        assert np.allclose(an[f, 72:79], golden[pre + "an_angles"], rtol=0, atol=2e-7)      # palmq goes through sinf/cosf
        assert np.array_equal(an[f, 79:84], golden[pre + "an_clenched"])


def test_no_cpu_fallback_symbols():
    """The product library must not link the oracle."""
    from hand_tracking_samples_amd import native
    import subprocess
    out = subprocess.run(["nm", "-D", native.lib_path()], capture_output=True, text=True).stdout
    assert "ho_" not in out
