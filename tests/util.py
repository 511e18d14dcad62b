"""Shared helpers for the test-suite (golden loading, deterministic inputs, comparisons)."""
import os

import numpy as np
import torch

from wavelet_monodepth_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
R18 = [64, 64, 128, 256, 512]
R50 = [64, 256, 512, 1024, 2048]


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def key_str(k):
    return k if isinstance(k, str) else "|".join(str(p) for p in k)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def sample(a, limit=4096):
    flat = np.asarray(a).reshape(-1)
    step = max(1, -(-flat.size // limit))
    return flat[::step]


def kitti_feats(batch, h, w, chans=R18, seed=1):
    return [t(f) for f in synth.encoder_features(batch, h, w, chans, seed=seed)]


def nyu_feats(batch, h, w, enc, seed=8, prefix="nyu_feat"):
    return [t(synth.normal((batch, c, h >> (k + 1), w >> (k + 1)), "%s%d" % (prefix, k), seed)) for k, c in enumerate(enc)]


def max_rel(a, b):
    """max |a-b| / max(|b|, 1e-6 * max|b|) -> a robust scalar relative error"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max() / scale)


def assert_close(a, b, rtol, what=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    err = max_rel(a, b)
    assert err <= rtol, "%s: max relative error %.3e > %.1e" % (what, err, rtol)


def assert_depth_close(disp_got, disp_ref, rtol, what="", min_depth=0.1, max_depth=100.0):
    """north_star's tolerance is "<= 1e-4 relative on depth maps": checked PER PIXEL on depth = 1 / (min_disp + (max_disp -
    min_disp) * disp) (KITTI/layers.py:16-25), where a norm-wise bound on the disparity is blind at small disparities."""
    to = lambda v: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)).astype(np.float64)
    lo, hi = 1.0 / max_depth, 1.0 / min_depth
    dg, dr = 1.0 / (lo + (hi - lo) * to(disp_got)), 1.0 / (lo + (hi - lo) * to(disp_ref))
    rel = float((np.abs(dg - dr) / np.abs(dr)).max())
    assert rel <= rtol, "%s: per-pixel relative depth error %.3e > %.1e" % (what, rel, rtol)


def photo_case(B=2, H=24, W=40, seed=31):
    """Two frames, a depth map, KITTI-like intrinsics and a small rigid motion — shared with the tests."""
    tgt = synth.uniform((B, 3, H, W), "ph_tgt", seed, 0.0, 1.0).astype(np.float32)
    src = synth.uniform((B, 3, H, W), "ph_src", seed, 0.0, 1.0).astype(np.float32)
    # smooth the frames a little so the bilinear sampling gradient is informative
    k = np.ones((3, 3), np.float32) / 9
    for a in (tgt, src):
        pad = np.pad(a, ((0, 0), (0, 0), (1, 1), (1, 1)), mode="edge")
        a[:] = sum(pad[:, :, i:i + H, j:j + W] * k[i, j] for i in range(3) for j in range(3))
    depth = synth.uniform((B, 1, H, W), "ph_depth", seed, 2.0, 30.0).astype(np.float32)
    K = np.tile(np.array([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32), (B, 1, 1))
    inv_K = np.linalg.inv(K).astype(np.float32)
    T = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    for b in range(B):
        ang = 0.02 * (b + 1)
        T[b, :3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
        T[b, :3, 3] = [0.3 * (b + 1), -0.05, 0.4]
    return tgt, src, depth, K, inv_K, T


def loss_case(B=2, H=32, W=64, seed=17, hints=False):
    """A synthetic minibatch for the photometric loss: colour pyramids of frames 0, -1, 1 (and "s"), intrinsics, poses,
    4 disparity scales.  -> (inputs, outputs) dicts of numpy arrays keyed like the reference trainer's."""
    from wavelet_monodepth_amd import synth
    frame_ids = [0, -1, 1] + (["s"] if hints else [])
    inputs, outputs = {}, {}
    k3 = np.ones((3, 3), np.float32) / 9
    for f in frame_ids:
        img = synth.uniform((B, 3, H, W), "lc_img%s" % f, seed, 0.0, 1.0).astype(np.float32)
        pad = np.pad(img, ((0, 0), (0, 0), (1, 1), (1, 1)), mode="edge")
        img = sum(pad[:, :, i:i + H, j:j + W] * k3[i, j] for i in range(3) for j in range(3)).astype(np.float32)
        for s in range(4):
            h, w = H >> s, W >> s
            inputs[("color", f, s)] = img.reshape(B, 3, h, 1 << s, w, 1 << s).mean((3, 5)).astype(np.float32)
    K = np.tile(np.array([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32), (B, 1, 1))
    inputs[("K", 0)], inputs[("inv_K", 0)] = K, np.linalg.inv(K).astype(np.float32)
    for f, tx in ((-1, -0.2), (1, 0.25)):
        T = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
        T[:, 0, 3], T[:, 2, 3] = tx, 0.1 * tx
        outputs[("cam_T_cam", 0, f)] = T
    st = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    st[:, 0, 3] = 0.1
    inputs["stereo_T"] = st
    for s in range(4):
        outputs[("disp", s)] = synth.uniform((B, 1, H >> s, W >> s), "lc_disp%d" % s, seed, 0.05, 0.9).astype(np.float32)
    if hints:
        inputs["depth_hint"] = synth.uniform((B, 1, H, W), "lc_hint", seed, 1.0, 40.0).astype(np.float32)
        inputs["depth_hint_mask"] = (synth.uniform((B, 1, H, W), "lc_hmask", seed, 0.0, 1.0) > 0.3).astype(np.float32)
    return inputs, outputs


def check_packed(out, gold, tol, exact=True, limit=4096):
    """Compare a decoder output dict with a PACKED full-size fixture (tests/golden/make_golden.py::pack_outputs): float
    maps as strided samples ("s|key"), boolean masks bit-packed ("m|key" + "mshape|key"), integers ("i|key")."""
    want = {k.split("|", 1)[1] for k in gold if k[:2] in ("s|", "m|", "i|")}
    have = {key_str(k) for k in out}
    assert want == have, want ^ have
    for k, v in out.items():
        ks = key_str(k)
        if torch.is_tensor(v) and v.dtype.is_floating_point:
            assert_close(sample(v.detach().cpu().numpy(), limit), gold["s|" + ks], tol, ks)
        elif torch.is_tensor(v):
            shape = tuple(int(n) for n in gold["mshape|" + ks])
            ref = np.unpackbits(gold["m|" + ks])[:int(np.prod(shape))].reshape(shape)
            assert tuple(v.shape) == shape, ks
            if exact:
                assert np.array_equal(v.cpu().numpy().astype(np.uint8), ref), ks
        elif exact:
            assert int(v) == int(gold["i|" + ks]), "%s: %d vs %d" % (ks, int(v), int(gold["i|" + ks]))


def unpack_mask(gold, key):
    shape = tuple(int(n) for n in gold["mshape|" + key])
    return np.unpackbits(gold["m|" + key])[:int(np.prod(shape))].reshape(shape)


def quarter_family_declines(name, C1, C2, masked=False, up=1, promise=1):
    """conv_wino32q_kernel (round 5) takes pure layers only -- every 8-channel chunk inside one source tensor, and an input mask
    over an upsampled operand only under the 2x2-constant promise (it has no generic gather); a forced launch on anything else
    must return WMD_ERR_UNSUPPORTED (-3) instead of computing something."""
    if not name.startswith("conv_wino32q"):
        return False
    return (C1 + C2) % 8 != 0 or (C2 > 0 and C1 % 8 != 0) or (masked and up == 2 and not promise)
