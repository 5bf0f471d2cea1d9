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
