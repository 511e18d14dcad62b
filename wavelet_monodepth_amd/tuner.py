"""Per-shape autotuning of the MFMA convolution (tile configuration x split-K), MIOpen-"find" style.

The C library is stateless: it exposes its configuration table (wmd_conv_num_configs/_config_name) and
lets the caller force a choice through wmd_conv_args.tune_cfg / tune_ksplit.  This module times the
candidates once per problem signature on the live GPU and remembers the winner for the process
(optionally across processes: WMD_TUNE_CACHE=/path/to/file.json).  WMD_AUTOTUNE=0 falls back to the
library's built-in cost model.
"""
import json
import os

import torch

enabled = os.environ.get("WMD_AUTOTUNE", "1") != "0"
_cache = {}
_cache_file = os.environ.get("WMD_TUNE_CACHE")
_loaded = False
KSPLITS = (1, 2, 3, 4, 5, 6, 8, 12, 16)
ranked = {}   # key -> [(config name, ksplit, ms)] of the last isolated sweep, best first (tools/step_tune.py starts from these)


def _load():
    global _loaded
    if _loaded:
        return
    _loaded = True
    if _cache_file and os.path.exists(_cache_file):
        with open(_cache_file) as f:
            for k, v in json.load(f).items():
                _cache[k] = tuple(v)


def _save():
    if _cache_file:
        with open(_cache_file, "w") as f:
            json.dump({k: list(v) for k, v in _cache.items()}, f, indent=0)


def preload(path):
    """Merge a committed cache file (read-only: nothing is written back to it)."""
    if os.path.exists(path):
        with open(path) as f:
            for k, v in json.load(f).items():
                _cache.setdefault(k, tuple(v))


def config_names():
    from . import _lib
    l = _lib.lib()
    return [l.wmd_conv_config_name(i).decode() for i in range(l.wmd_conv_num_configs())]


_names = None


def _index_of(name):
    global _names
    if _names is None:
        _names = config_names()
    return _names.index(name) + 1 if name in _names else None


def lookup(key):
    """-> (cfg1, ksplit) or None.  The cache stores configuration NAMES, so it survives edits of the table."""
    _load()
    hit = _cache.get(key)
    if hit is None:
        return None
    if hit[0] == "":
        return (0, hit[1])
    idx = _index_of(hit[0])
    return None if idx is None else (idx, hit[1])


def _time(launch, cfg, ks, reps, e0, e1):
    t = float("inf")
    for _ in range(reps):
        e0.record()
        launch(cfg, ks)
        e1.record()
        e1.synchronize()
        t = min(t, e0.elapsed_time(e1))
    return t


def tune(key, taps, launch):
    """launch(cfg1, ks) -> status int (0 ok). Returns (cfg1, ks) with the smallest GPU time:
    a coarse sweep (min of 2 runs each) followed by a 6-run play-off between the five best."""
    names = config_names()
    cands = [i + 1 for i, n in enumerate(names) if n.endswith(",%d>" % taps) or (taps == 9 and n.startswith("conv_wino"))]
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    results = []
    best_t = float("inf")
    for cfg in cands:
        # k > 0: k slices finished inside the convolution where the kernel can (round 6); -k: the same slices summed by the
        # second-stage kernel -- the planner refuses -k for kernels that have only that form (k then already means it)
        for ks in KSPLITS + tuple(-k for k in KSPLITS if k > 1):
            if launch(cfg, ks) != 0:      # invalid combination for this shape (planner refuses)
                continue
            t = _time(launch, cfg, ks, 2, e0, e1)
            results.append((t, cfg, ks))
            best_t = min(best_t, t)
            if ks == 1 and t > 4.0 * best_t:
                break  # hopeless tile shape for this problem: do not sweep its splits
    if not results:
        _cache[key] = ("", 0)
        return (0, 0)
    results.sort()
    ranked[key] = [(names[cfg - 1], ks, t) for t, cfg, ks in results[:12]]
    final = sorted((_time(launch, cfg, ks, 6, e0, e1), cfg, ks) for _, cfg, ks in results[:5])
    t, cfg, ks = final[0]
    _cache[key] = (names[cfg - 1], ks)
    _save()
    if os.environ.get("WMD_TUNE_VERBOSE"):
        print("[wmd tuner] %s -> %s ksplit %d (%.1f us)" % (key, names[cfg - 1], ks, t * 1e3))
    return (cfg, ks)


def choose(key, cands, launch):
    """Generic form for the other kernel families (weight gradient): cands = [(label, arg), ...], launch(arg) -> status.
    Returns the arg of the fastest candidate (min of 2 runs each, then a 5-run play-off between the best three); the
    winner's LABEL is cached under `key`, so the cache survives table edits.  With tuning disabled or inside a stream
    capture the first candidate (the library's own model) is returned."""
    _load()
    hit = _cache.get(key)
    if hit is not None:
        for label, arg in cands:
            if label == hit[0]:
                return arg
    if not enabled or torch.cuda.is_current_stream_capturing() or len(cands) == 1:
        return cands[0][1]
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)

    def t_of(arg, reps):
        best = float("inf")
        for _ in range(reps):
            e0.record()
            st = launch(arg)
            e1.record()
            e1.synchronize()
            if st != 0:
                return None
            best = min(best, e0.elapsed_time(e1))
        return best

    res = []
    for label, arg in cands:
        if launch(arg) != 0:          # warm-up run doubles as the validity check
            continue
        t = t_of(arg, 2)
        if t is not None:
            res.append((t, label, arg))
    if not res:
        return cands[0][1]
    res.sort(key=lambda r: r[0])
    final = sorted(((t_of(arg, 5), label, arg) for _, label, arg in res[:3]), key=lambda r: r[0])
    t, label, arg = final[0]
    _cache[key] = (label, 0)
    _save()
    if os.environ.get("WMD_TUNE_VERBOSE"):
        print("[wmd tuner] %s -> %s (%.1f us)" % (key, label, t * 1e3))
    return arg
