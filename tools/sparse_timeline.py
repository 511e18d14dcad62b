"""Timeline of one hipGraph replay of the batch-1 sparse (or dense) KITTI decoder from a rocprofv3 kernel trace (development aid).
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $REPO/tools/sparse_timeline.py run [thresh] [dense] [batch=B] [density=p | contour=p3,p2,p1]
    python tools/sparse_timeline.py parse /tmp/tl/*/*kernel_trace.csv
run: 3 warm-ups, then 6 replays separated by a device synchronisation; parse: the last replay as (start offset, duration, gap, kernel)."""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import numpy as np
    import torch
    from wavelet_monodepth_amd import synth
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder, SparseDepthWaveProgressiveDecoder
    dev = torch.device("cuda:0")
    chans = [64, 64, 128, 256, 512]
    dense = "dense" in sys.argv
    thr = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "dense" else 0.15
    cls = DepthWaveProgressiveDecoder if dense else SparseDepthWaveProgressiveDecoder
    dec = synth.fill_state_dict(cls(np.array(chans)), seed=1).to(dev)
    dec.enable_graph(True)
    opt = lambda k, d: next((a.split("=")[1] for a in sys.argv if a.startswith(k + "=")), d)
    B, p = int(opt("batch", "1")), float(opt("density", "0"))
    feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(B, 192, 640, chans, seed=1)]
    force = None
    if p > 0:      # controlled density: the same injected masks as tools/config_bench.py
        force = {i: (torch.from_numpy(synth.uniform((h, w), "dens%d" % i, 3, 0.0, 1.0)) < p).to(torch.uint8).to(dev)
                 for i, (h, w) in zip((3, 2, 1), ((12, 40), (24, 80), (48, 160)))}
    cont = opt("contour", "")
    if cont:       # contour=p3,p2,p1: outline-shaped masks (synth.contour_mask), one density per level
        ps = [float(v) for v in cont.split(",")]
        force = {i: torch.from_numpy(np.stack([synth.contour_mask(h, w, pp, "contour", 10 + k) for k in range(B)])).to(dev)
                 for (i, (h, w)), pp in zip(((3, (12, 40)), (2, (24, 80)), (1, (48, 160))), ps)}
    with torch.no_grad():
        for _ in range(9):
            dec(feats) if dense else dec(feats, thr, _force_masks=force)
            torch.cuda.synchronize()
else:
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "wmd" in r["Kernel_Name"] or "PackMany" in r["Kernel_Name"] or "at::" in r["Kernel_Name"]
            or "rocclr" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # replays are separated by a host synchronisation: split at gaps > 50 us, keep the last group
    groups, cur = [], []
    for r in rows:
        if cur and int(r["Start_Timestamp"]) - int(cur[-1]["End_Timestamp"]) > 50000:
            groups.append(cur)
            cur = []
        cur.append(r)
    groups.append(cur)
    g = groups[-1]
    t0 = int(g[0]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    for r in g:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        m = re.search(r"(?:wmd::)?([A-Za-z0-9_]+)(<[^>]*>)?\(", r["Kernel_Name"])
        name = (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else r["Kernel_Name"][:60]
        print("%8.1f us  dur %6.1f  gap %5.1f  grid %8s wg %4s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), name[:70]))
        busy += e - s
        prev_end = e
    print("replay: %d kernels, %.1f us end to end, %.1f us busy (%d replay groups seen)" % (len(g), (prev_end - t0) / 1e3, busy / 1e3, len(groups)))
