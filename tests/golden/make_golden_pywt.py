"""Independent Haar pins from PyWavelets 1.1.1 (numpy only).  Run in the build container with
    /opt/conda/bin/python3.9 tests/golden/make_golden_pywt.py
pytorch_wavelets takes its filters from pywt and implements idwt2/dwt2 semantics; (LH,HL,HH) map to
pywt's (cH, cV, cD).  Inputs come from wavelet_monodepth_amd/synth.py (loaded by path: no torch)."""
import importlib.util
import os

import numpy as np
import pywt

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("synth", os.path.join(HERE, "..", "..", "wavelet_monodepth_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)

out = {"pywt_version": np.asarray([int(v) for v in pywt.__version__.split(".")])}
for name, (h, w) in {"a": (4, 6), "b": (12, 40), "c": (7, 5), "d": (24, 80)}.items():
    yl = synth.normal((h, w), "pywt_yl_" + name, 11).astype(np.float64)
    yh = synth.normal((3, h, w), "pywt_yh_" + name, 11).astype(np.float64)
    rec = pywt.idwt2((yl, (yh[0], yh[1], yh[2])), "haar", mode="zero")
    out["idwt_" + name] = rec
# "o*": odd sizes -- mode="reflect" extends an odd axis by one reflected sample (round 2)
for name, (h, w, J) in {"a": (8, 12, 1), "b": (48, 160, 4), "c": (240, 320, 4), "o1": (7, 9, 1), "o2": (15, 22, 3), "o3": (30, 45, 4)}.items():
    x = synth.normal((h, w), "pywt_x_" + name, 12).astype(np.float64)
    ll = x
    for j in range(J):
        ll, (cH, cV, cD) = pywt.dwt2(ll, "haar", mode="reflect")
        if h * w <= 48 * 160 or j == J - 1:
            out["dwt_%s_yh%d" % (name, j)] = np.stack([cH, cV, cD])
    out["dwt_%s_yl" % name] = ll
np.savez_compressed(os.path.join(HERE, "pywt_haar.npz"), **out)
print("pywt goldens written:", sorted(out))
