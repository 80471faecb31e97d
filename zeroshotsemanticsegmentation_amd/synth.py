"""Deterministic synthetic weights / inputs (no network: no VGG16 download, no PASCAL images).

Everything is produced by one counter-based generator (splitmix64 of (stream id, element index)) so that
this container (golden capture from the reference), the CPU oracle and the GPU box all see identical
numbers without depending on any torch / numpy RNG implementation.

Synthetic-input recipe: images uint8 U{0..255} -> BGR - mean_bgr -> f32 NCHW
(context_dataset.py:51,143-148); labels piecewise constant on 32x32 blocks, 5 % of the pixels -1;
embeddings: the reference K x E matrices (tests/golden/embeddings_*.npy) or a synthetic 59 x 300 matrix.
"""
import math

import numpy as np

MEAN_BGR = np.array([104.00698793, 116.66876762, 122.67891434], dtype=np.float64)  # context_dataset.py:51

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform01(stream, n, offset=0):
    """n doubles in [0,1) with 53 random bits; element i depends only on (stream, offset + i)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        key = _splitmix64(np.uint64(stream) * np.uint64(0xD1342543DE82EF95) + np.uint64(0x2545F4914F6CDD1D))
        h = _splitmix64(idx ^ key)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(stream, shape, lo, hi):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(stream, n)).astype(np.float32).reshape(shape)


# (name, Cout, Cin, K) in forward order -- models.py:43-98
CONV_LAYERS = [
    ("conv1_1", 64, 3, 3), ("conv1_2", 64, 64, 3),
    ("conv2_1", 128, 64, 3), ("conv2_2", 128, 128, 3),
    ("conv3_1", 256, 128, 3), ("conv3_2", 256, 256, 3), ("conv3_3", 256, 256, 3),
    ("conv4_1", 512, 256, 3), ("conv4_2", 512, 512, 3), ("conv4_3", 512, 512, 3),
    ("conv5_1", 512, 512, 3), ("conv5_2", 512, 512, 3), ("conv5_3", 512, 512, 3),
    ("fc6", 4096, 512, 7), ("fc7", 4096, 4096, 1),
]


def layer_table(n_class, width_div=1):
    """[(name, Cout, Cin, K)] incl. the two heads. width_div > 1 shrinks every hidden width (tests only)."""
    d = width_div
    t = [(n, max(co // d, 8), (ci if ci == 3 else max(ci // d, 8)), k) for n, co, ci, k in CONV_LAYERS]
    hidden = t[-1][1]
    t.append(("score_fr", n_class, hidden, 1))
    t.append(("seenmask_score", 2, hidden, 1))
    return t


def make_params(n_class, seed=1337, width_div=1):
    """dict name -> float32 array in TORCH layout (O,I,KH,KW) / (O,): He-uniform weights, small biases.

    The two ConvTranspose2d weights (upscore, seenmask_upscore) are the fixed bilinear kernels of
    models.py:11-24 and are not part of this dict (see bilinear_weight)."""
    out = {}
    for li, (name, co, ci, k) in enumerate(layer_table(n_class, width_div)):
        fan_in = ci * k * k
        b = math.sqrt(6.0 / fan_in)
        out[name + ".weight"] = uniform(seed * 1000 + 2 * li, (co, ci, k, k), -b, b)
        out[name + ".bias"] = uniform(seed * 1000 + 2 * li + 1, (co,), -0.1, 0.1)
    return out


def bilinear_filter_1d(k=64):
    """1-D factor of get_upsampling_weight (models.py:13-20), float64."""
    factor = (k + 1) // 2
    center = factor - 1 if k % 2 == 1 else factor - 0.5
    t = np.arange(k, dtype=np.float64)
    return 1.0 - np.abs(t - center) / factor


def bilinear_weight(cin, cout, k=64):
    """(cin, cout, k, k) float32: the filter on the channel diagonal, zeros elsewhere (models.py:21-24)."""
    f = bilinear_filter_1d(k)
    filt = (f[:, None] * f[None, :]).astype(np.float32)      # float64 product, then float32 (models.py:24)
    w = np.zeros((cin, cout, k, k), dtype=np.float32)
    w[range(min(cin, cout)), range(min(cin, cout))] = filt
    return w


def make_images(B, H, W, seed=1337):
    """(B,3,H,W) float32: uint8 RGB noise -> BGR -> minus mean_bgr (context_dataset.py:143-148)."""
    u = np.floor(uniform01(seed * 1000 + 900, B * H * W * 3) * 256.0).reshape(B, H, W, 3)
    bgr = u[..., ::-1] - MEAN_BGR
    return np.ascontiguousarray(bgr.transpose(0, 3, 1, 2)).astype(np.float32)


def make_labels(B, H, W, K, seed=1337, block=32, ignore_frac=0.05, classes=None):
    """(B,H,W) int64: block-constant classes drawn from `classes` (default range(K)), 5 % set to -1."""
    classes = np.arange(K) if classes is None else np.asarray(classes)
    hb, wb = (H + block - 1) // block, (W + block - 1) // block
    pick = np.floor(uniform01(seed * 1000 + 901, B * hb * wb) * len(classes)).astype(np.int64).reshape(B, hb, wb)
    lbl = classes[pick].repeat(block, axis=1).repeat(block, axis=2)[:, :H, :W].copy()
    ign = uniform01(seed * 1000 + 902, B * H * W).reshape(B, H, W) < ignore_frac
    lbl[ign] = -1
    return lbl.astype(np.int64)


def make_embeddings(K=59, E=300, seed=1337):
    """Synthetic K x E class-embedding matrix with the row-norm profile of the reference's 300-d files
    (norms in [0.63, 1.0], exactly one row of norm 1): SURVEY.md section 8-d recipe."""
    g = uniform01(seed * 1000 + 903, 2 * K * E).reshape(2, K, E)
    # Box-Muller on the counter stream
    z = np.sqrt(-2.0 * np.log(1.0 - g[0])) * np.cos(2.0 * np.pi * g[1])
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    r = 0.63 + 0.37 * uniform01(seed * 1000 + 904, K)
    r[int(np.argmax(r))] = 1.0
    return (z * r[:, None]).astype(np.float32)


def unseen_bits(unseen):
    bits = 0
    for k in unseen:
        bits |= 1 << int(k)
    return bits
