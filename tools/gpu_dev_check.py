"""Development aid: run every forward parity case and print the error instead of stopping at the
first failure (one gpurun call = one full picture)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

from oracle import decoder_ref as R
from wavelet_monodepth_amd import synth, ops
from util import R18, key_str, kitti_feats, load_golden, max_rel, t
import test_gpu_parity as T

dev = torch.device("cuda:0")
print("device:", torch.cuda.get_device_name(0))

for case in T.CONV_CASES:
    try:
        T.test_conv_forward(dev, case)
        print("conv OK  ", case)
    except Exception as e:
        print("conv FAIL", case, str(e).splitlines()[-1][:200])

for shape in [(2, 1, 6, 20), (12, 1, 96, 320), (1, 1, 5, 7)]:
    try:
        T.test_idwt_forward(dev, shape)
        print("idwt OK  ", shape)
    except Exception as e:
        print("idwt FAIL", shape, str(e)[:200])

for fn in (T.test_idwt_vs_pywavelets_golden, T.test_dwt_vs_pywavelets_golden_and_roundtrip, T.test_idwt_backward,
           T.test_kitti_dense_decoder_vs_reference_golden, T.test_kitti_dense_decoder_config2_vs_oracle,
           T.test_kitti_dense_decoder_batch12_properties, T.test_kitti_baseline_decoder_vs_reference_golden):
    try:
        fn(dev)
        print("OK  ", fn.__name__)
    except Exception as e:
        print("FAIL", fn.__name__, str(e).splitlines()[-1][:300])
        traceback.print_exc(limit=2)

# quick timing of the config-2 forward
from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=1).to(dev)
feats = [f.to(dev) for f in kitti_feats(12, 192, 640)]
with torch.no_grad():
    for _ in range(3):
        dec(feats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        dec(feats)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print("config2 forward: %.3f ms/batch -> %.1f frames/s" % (dt * 1e3, 12 / dt))
