"""Seeded synthetic CNN weights in the reference's `.cnnb` layout.

The trained `assets/handposedd.cnnb` is not shipped with the reference, so benches and parity tests use
weights drawn from a counter-based splitmix64 stream in the Xavier range the reference's own `init()` uses
(third_party/cnn.h:282,448).  Layout (cnn.h:288,454,590): conv1 W[400] B[16]; conv2 W[16384] B[64];
fc1 W[2304*2048] B[2048]; fc2 W[2048*2304] B[2304] -- raw little-endian fp32, 9 458 400 values.

The same generator is restated in oracle/ref_harness.cpp so the reference CNN can be loaded with identical bits.

`side=128` gives the weights of the 128x128-input variant of the same net (BASELINE configs[4], SURVEY 8d "config 5 (ii)":
conv 5x5 -> 124, 2 pools -> 31, conv 4x4 -> 28, pool -> 14, FC 12544 -> 2048 -> 2304): same stream, same ranges, only the first
fully connected layer is larger (30 429 920 values).
"""
import numpy as np

CNNB_COUNT = 9458400
DEFAULT_SEED = 0x5EED0001
DEFAULT_FC2_GAIN = 24.0      # makes the softmax heat-maps peaky, so arg-max decoding is well conditioned



def features(side=64):
    """inputs of the first fully connected layer: 64 channels x (pooled side)^2 -- 2304 for a 64x64 input, 12544 for 128x128"""
    p = (((side - 4) // 4) - 3) // 2
    return 64 * p * p


def _layers(side=64):  # (n_weights, n_bias, fan_in + fan_out)
    f = features(side)
    return ((400, 16, 25.0 * 1 + 25.0 * 16),
            (16384, 64, 16.0 * 16 + 16.0 * 64),
            (f * 2048, 2048, float(f) + 2048.0),
            (2048 * 2304, 2304, 2048.0 + 2304.0))


def cnnb_count(side=64):
    return sum(nw + nb for nw, nb, _ in _layers(side))



def _splitmix64(seed, idx):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def make_cnnb(seed=DEFAULT_SEED, fc2_gain=DEFAULT_FC2_GAIN, side=64):
    """Return the flat fp32 weight vector (len CNNB_COUNT for side 64)."""
    out = np.empty(cnnb_count(side), dtype=np.float32)
    ctr = 0
    pos = 0
    for li, (nw, nb, fan) in enumerate(_layers(side)):
        gain = fc2_gain if li == 3 else 1.0
        for n, rng in ((nw, np.sqrt(6.0 / fan) * gain), (nb, 0.05 * gain)):
            idx = np.arange(ctr, ctr + n, dtype=np.uint64)
            u = (_splitmix64(seed, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
            out[pos:pos + n] = ((2.0 * u - 1.0) * rng).astype(np.float32)
            ctr += n
            pos += n
    return out


def make_cnnb128(seed=DEFAULT_SEED, fc2_gain=DEFAULT_FC2_GAIN):
    return make_cnnb(seed, fc2_gain, side=128)


def load_cnnb(path):
    """Read a `.cnnb` weight file (what CNN::loadb consumes, cnn.h:590-592)."""
    w = np.fromfile(path, dtype="<f4")
    if w.size != CNNB_COUNT:
        raise ValueError("%s: expected %d fp32 values, found %d" % (path, CNNB_COUNT, w.size))
    return w


def save_cnnb(path, w):
    np.asarray(w, dtype="<f4").tofile(path)
