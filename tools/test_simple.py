"""Counterpart of the reference's KITTI/test_simple.py (BASELINE.json configs[0], plumbing): one image through
encoder + (dense|sparse) wavelet decoder, writing the same artefacts the reference writes
(/root/reference/KITTI/test_simple.py:138-164): `<name>_disp.npy` (scaled disparity), one `.npy` per wavelet
plane, and the depth range used.  The reference runs this on CPU; this package has no CPU path by design, so it
needs the MI355X.  Without --image a synthetic 640x192 picture is generated; without --weights the networks are
randomly initialised (no released checkpoints can be fetched here).

    python tools/test_simple.py --out /tmp/out [--sparse --threshold 0.05] [--weights DIR] [--image FILE.npy]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from wavelet_monodepth_amd import synth
from wavelet_monodepth_amd.encoders import ResnetEncoder
from wavelet_monodepth_amd.kitti import make_depth_decoder


def disp_to_depth(disp, min_depth, max_depth):   # KITTI/layers.py:16-25
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled = min_disp + (max_disp - min_disp) * disp
    return scaled, 1 / scaled


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--image", help=".npy float array [3,H,W] in [0,1]")
    ap.add_argument("--weights", help="folder with encoder.pth / depth.pth")
    ap.add_argument("--num_layers", type=int, default=18)
    ap.add_argument("--sparse", action="store_true")
    ap.add_argument("--threshold", type=float, default=0.05)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    os.makedirs(args.out, exist_ok=True)
    enc = ResnetEncoder(args.num_layers)
    dec = make_depth_decoder(enc.num_ch_enc, range(4), use_wavelets=True, use_sparse=args.sparse)
    if args.weights:
        sd = torch.load(os.path.join(args.weights, "encoder.pth"), map_location="cpu")
        enc.load_state_dict({k: v for k, v in sd.items() if k in enc.state_dict()})     # test_simple.py:82-88
        dec.load_state_dict(torch.load(os.path.join(args.weights, "depth.pth"), map_location="cpu"))  # strict, :101-102
    else:
        synth.fill_state_dict(dec, seed=1)
    enc.to(dev).eval()
    dec.to(dev).eval()
    if args.image:
        img = torch.from_numpy(np.load(args.image)).float()[None]
    else:
        img = torch.from_numpy(synth.uniform((1, 3, args.height, args.width), "image", 0, 0.0, 1.0))
    with torch.no_grad():
        feats = enc(img.to(dev))
        out = dec(feats, args.threshold) if args.sparse else dec(feats)
    scaled, depth = disp_to_depth(out[("disp", 0)], 0.1, 100)
    np.save(os.path.join(args.out, "image_disp.npy"), scaled.cpu().numpy())
    for s in range(4):
        for band in ("LL", "LH", "HL", "HH"):
            np.save(os.path.join(args.out, "image_wavelets_%d_%s.npy" % (s, band)), out[("wavelets", s, band)].cpu().numpy())
    print("wrote %d files to %s; depth range %.3f .. %.3f m%s" % (17, args.out, float(depth.min()), float(depth.max()),
          ("; total_ops %d" % out["total_ops"]) if args.sparse else ""))


if __name__ == "__main__":
    main()
