"""CNN training step and label synthesis: CNN::Train (cnn.h:558-580) with the labels of GatherHandExpectedCNN
(handtrack.h:160-173).  tests/golden/train3.htfx (reference): 3 frames x 2 epochs of SGD from the seeded weights, step 0.001."""
import ctypes as C
import os

import numpy as np
import pytest

import htfx
import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
G = htfx.load(os.path.join(HERE, "golden", "train3.htfx"))
OFF = {"W1": 0, "B1": 400, "W2": 416, "B2": 16800, "W3": 16864, "B3": 16864 + 2304 * 2048, "W4": 16864 + 2304 * 2048 + 2048, "B4": 16864 + 2304 * 2048 + 2048 + 2048 * 2304}


def _hcam(cam12):
    c = np.array(cam12, np.float32); c[:4] = c[:4] / np.float32(4.0)      # camsub(cam, 4) misc_image.h:60
    return ol.camera(c, 16, 16)


def _inputs():
    L = ol.lib()
    xs = []
    for f in range(3):
        x = np.zeros(4096, np.float32)
        L.ho_cnn_input(ol.u16ptr(np.ascontiguousarray(G["f%d/depth" % f].reshape(-1))), 4096, float(G["f%d/cam" % f][4]), 0.1, 0.7, ol.fptr(x))
        xs.append(x)
    return xs


@pytest.mark.parametrize("f", range(3))
def test_labels_match_reference(f):
    exp = np.zeros(2304, np.float32); vals = np.zeros(16, np.float32)
    cam = _hcam(G["f%d/cam" % f])
    ol.lib().ho_expected_cnn(ol.fptr(np.ascontiguousarray(G["f%d/pose" % f])), C.byref(cam), ol.fptr(exp), ol.fptr(vals))
    assert np.array_equal(vals, G["f%d/vals" % f])
    assert np.array_equal(exp, G["f%d/labels" % f])


def check_weights(w, tol=0.0):
    def close(a, b):
        return np.array_equal(a, b) if tol == 0.0 else np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())
    assert close(w[OFF["W1"]:OFF["W1"] + 400], G["W1"]) and close(w[OFF["B1"]:OFF["B1"] + 16], G["B1"])
    assert close(w[OFF["W2"]:OFF["W2"] + 1024], G["W2_head"]) and close(w[OFF["B2"]:OFF["B2"] + 64], G["B2"])
    assert close(w[OFF["B3"]:OFF["B3"] + 2048], G["B3"]) and close(w[OFF["B4"]:OFF["B4"] + 2304], G["B4"])
    assert close(w[OFF["W3"]:OFF["B3"]][::9973], G["W3_every9973"]) and close(w[OFF["W4"]:OFF["B4"]][::9973], G["W4_every9973"])


def test_oracle_training_matches_reference(weights):
    w = np.array(weights, np.float32, copy=True)
    xs = _inputs()
    L = ol.lib()
    mse = []
    for e in range(2):
        for f in range(3):
            mse.append(L.ho_cnn_train(ol.fptr(w), ol.fptr(xs[f]), ol.fptr(np.ascontiguousarray(G["f%d/labels" % f])), 0.001))
    assert np.array_equal(np.array(mse, np.float32), G["mse"])
    check_weights(w)
    y = np.zeros(2304, np.float32)
    L.ho_cnn_eval(ol.fptr(w), ol.fptr(xs[0]), ol.fptr(y), None)
    assert np.array_equal(y, G["eval0_after"])


@pytest.mark.parametrize("f", range(3))
def test_library_labels_match_reference(f):
    from hand_tracking_samples_amd import native
    assert np.array_equal(native.expected_cnn(G["f%d/pose" % f], G["f%d/cam" % f]), G["f%d/labels" % f])


@pytest.mark.gpu
def test_gpu_training_matches_reference(weights):
    """6 SGD steps on the device against the reference: the long fully connected dot products are reduced in parallel, so agreement is to
    float rounding (1e-5 relative), not bit for bit."""
    from hand_tracking_samples_amd import native
    ctx = native.Context(ol.MODEL, 1)
    ctx.load_weights(weights)
    xs = np.stack(_inputs()); ts = np.stack([G["f%d/labels" % f] for f in range(3)])
    mse = np.concatenate([ctx.cnn_train(xs, ts, 0.001) for _ in range(2)])
    print("mse", mse, "reference", G["mse"])
    assert np.abs(mse - G["mse"]).max() <= 1e-5 * np.abs(G["mse"]).max()
    check_weights(ctx.cnn_get_weights(), tol=1e-5)
    y = ctx.cnn_eval(xs[0:1])[0]      # the inference kernels see the trained weights (conv2 repacked)
    assert np.abs(y - G["eval0_after"]).max() <= 2e-5
    ctx.close()
