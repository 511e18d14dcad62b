"""Deterministic, RNG-library-independent synthetic tensors.

Every test fixture, the smoke test and bench.py draw inputs and weights from here, so golden files
only need to store *outputs*: the same (name, shape, seed) always yields bit-identical fp32 data on
any machine (pure integer hashing + exact float arithmetic, no transcendental functions).
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _seed_of(name, seed):
    return np.uint64((zlib.crc32(name.encode()) << 20) ^ (int(seed) * 0x9E3779B1 & 0xFFFFFFFFFFFF))


def uniform(shape, name="x", seed=0, lo=-1.0, hi=1.0):
    """fp32 array, i.i.d. uniform in [lo, hi) with 24-bit resolution."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + _seed_of(name, seed))
    u = (h >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(shape, name="x", seed=0, std=1.0):
    """fp32 array, approximately N(0, std^2): centred sum of four 24-bit uniforms (Irwin-Hall)."""
    acc = np.zeros(int(np.prod(shape)) if len(shape) else 1, dtype=np.float64)
    for k in range(4):
        acc += uniform((acc.size,), name + "#%d" % k, seed, 0.0, 1.0).astype(np.float64)
    # var of the sum of 4 U(0,1) = 4/12
    return ((acc - 2.0) * (std / np.sqrt(4.0 / 12.0))).astype(np.float32).reshape(shape)


def conv_params(name, cout, cin, k, seed=0):
    """Weight/bias with PyTorch's default Conv2d scale (bound = 1/sqrt(fan_in))."""
    bound = 1.0 / np.sqrt(cin * k * k)
    w = uniform((cout, cin, k, k), name + ".weight", seed, -bound, bound)
    b = uniform((cout,), name + ".bias", seed, -bound, bound)
    return w, b


def fill_state_dict(module, seed=0):
    """Overwrite every floating parameter of `module` (in state_dict order) with synth values.
    Conv weights/biases get the 1/sqrt(fan_in) scale; registered buffers are left untouched."""
    import torch

    params = dict(module.named_parameters())
    with torch.no_grad():
        for name, p in params.items():
            if name.endswith(".weight") and p.dim() == 4:
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
            elif name.endswith(".bias"):
                w = params.get(name[: -len(".bias")] + ".weight")
                fan_in = (w.shape[1] * w.shape[2] * w.shape[3]) if w is not None and w.dim() == 4 else p.numel()
            else:
                fan_in = p.numel()
            bound = 1.0 / np.sqrt(max(fan_in, 1))
            p.copy_(torch.from_numpy(uniform(tuple(p.shape), name, seed, -bound, bound)))
    return module


def encoder_features(batch, height, width, num_ch_enc, seed=0, dist="normal"):
    """Five synthetic encoder feature maps at strides 2,4,8,16,32."""
    feats = []
    for k, c in enumerate(num_ch_enc):
        shape = (batch, int(c), height >> (k + 1), width >> (k + 1))
        if dist == "normal":
            feats.append(normal(shape, "feat%d" % k, seed))
        else:
            feats.append(uniform(shape, "feat%d" % k, seed))
    return feats


def contour_mask(h, w, density, name="contour", seed=0):
    """uint8 [h,w] mask of thin level-set contours of a smooth random field covering ~`density` of the grid: the shape the
    decoders' wavelet masks have on photographs (depth discontinuities along object outlines), as opposed to i.i.d. pixels
    (`uniform(...) < density`), whose dilations cover almost everything.  The same (name, seed) at another resolution samples
    the same field, so the masks of successive decoder levels trace the same outlines."""
    ph = uniform((6, 4), name, seed, 0.0, 1.0).astype(np.float64)
    ys = (np.arange(h, dtype=np.float64) + 0.5) / h
    xs = (np.arange(w, dtype=np.float64) + 0.5) / h          # same units as ys: the field is isotropic
    f = np.zeros((h, w), dtype=np.float64)
    for a, fy, fx, p in ph:
        f += (0.5 + a) * np.sin(2 * np.pi * ((fy - 0.5) * 1.6 * ys[:, None] + (fx - 0.5) * 1.6 * xs[None, :] + p))
    d = np.abs(f - np.median(f))
    eps = np.quantile(d, min(max(float(density), 0.0), 1.0))
    return (d <= eps).astype(np.uint8)
