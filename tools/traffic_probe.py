"""Round 6: what do the coarse split-K layers fetch?  Launches ONE trunk layer of BASELINE config 2 (batch 12) a few
times with a forced kernel / K split / finish form, nothing else on the device -- to be run under `rocprofv3 --pmc <TCC counters>`
(tools/traffic_session.sh), one counter group per pass.
    python tools/traffic_probe.py <layer> <ksplit>      layer in L0 L1 L2 L14;  ksplit: k (in-kernel finish), -k (second-stage kernel), 1
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wavelet_monodepth_amd import _lib, ops, synth, tuner

LAYERS = {  # name: (B, H, W, C1, up, C2, Cout, kernel)
    "L0": (12, 6, 20, 512, 1, 0, 256, "conv_wino32q_kernel<4,32,8>"),
    "L1": (12, 12, 40, 256, 2, 256, 256, "conv_wino32_kernel<6,40,2,8>"),
    "L2": (12, 12, 40, 256, 1, 0, 128, "conv_wino32_kernel<12,40,4,8>"),
    "L14": (12, 96, 320, 32, 2, 64, 32, "conv_wino32_kernel<8,32,2,8>"),
}
layer, ks = sys.argv[1], int(sys.argv[2])
B, H, W, C1, up, C2, Cout, name = LAYERS[layer]
dev = torch.device("cuda:0")
t = torch.from_numpy
x1 = t(synth.normal((B, C1, H // up, W // up), "px1", 1)).to(dev)
x2 = t(synth.normal((B, C2, H, W), "px2", 1)).to(dev) if C2 else None
w, b = [t(a).to(dev) for a in synth.conv_params("pw", Cout, C1 + C2, 3, 1)]
wp, ww = ops.pack_weights(w), ops.pack_weights_wino(w)
names = tuner.config_names()
l = _lib.lib()
y = torch.empty((B, Cout, H, W), device=dev)
a = _lib.ConvArgs(B=B, H=H, W=W, C1=C1, up1=up, C2=C2, Cout=Cout, ksize=3, pad_mode=1, act=1, slope=0.0, x1=x1.data_ptr(),
                  x2=None if x2 is None else x2.data_ptr(), wp=wp.data_ptr(), bias=b.data_ptr(), y=y.data_ptr(), workspace=None,
                  workspace_floats=0, tune_cfg=names.index(name) + 1, tune_ksplit=ks, wp_wino=ww.data_ptr())
n = l.wmd_conv_fwd_workspace_floats(C.byref(a))
ws = torch.empty(max(n, 1), device=dev)
a.workspace, a.workspace_floats = ws.data_ptr(), n
# something else between the launches evicts the layer's lines from the caches, as the rest of the decoder does in a real step
spoil = torch.empty(96 * 1024 * 1024, device=dev)      # 384 MB > the 256 MB Infinity Cache
for rep in range(6):
    if os.environ.get("WMD_PROBE_SPOIL", "1") == "1":
        spoil.fill_(float(rep))
    _lib.check(l.wmd_conv_fwd(C.byref(a), torch.cuda.current_stream().cuda_stream), name)
torch.cuda.synchronize()
alg = 4.0 * (x1.numel() + (x2.numel() if x2 is not None else 0) + w.numel() * 16 / 9 + y.numel())
print("layer %s %s ks %d: inputs %.2f MB, Winograd weights %.2f MB, output %.2f MB, partial planes %.2f MB" % (
    layer, name, ks, 4e-6 * (x1.numel() + (x2.numel() if x2 is not None else 0)), 4e-6 * w.numel() * 16 / 9, 4e-6 * y.numel(), 4e-6 * n))
