"""Operator layer of the threshold-gated sparse path (batch 1, inference only; the reference forbids
training with it, KITTI/trainer.py:35-36).  Thin wrappers over the wmd_mask_* / wmd_sparse_conv entry points;
see include/wmd.h for the layout decision (dense zero-initialised activations + masks + compacted pixel lists).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT, PAD, check, current_stream, ptr


def _on_gpu(*tensors):
    """No CPU implementation in the package: every tensor handed to the library must live on the GPU."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.WmdError("wavelet_monodepth_amd sparse ops run on the MI355X only (got a %s tensor)" % t.device)


def minmax(x):
    """-> device tensor [min, max] of x (depth_decoder.py:308)."""
    _on_gpu(x)
    out = torch.empty(2, device=x.device, dtype=torch.float32)
    x = x.contiguous()
    check(_lib.lib().wmd_minmax(ptr(x), x.numel(), ptr(out), current_stream()), "wmd_minmax")
    return out


def mask_threshold(yh, mm, thresh_ratio):
    """yh [1,1,3,h,w] (or [3,h,w]) -> uint8 mask [h,w]: max_b |yh_b| > (max-min)*ratio (:308-309)."""
    _on_gpu(yh, mm)
    h, w = yh.shape[-2:]
    yh = yh.contiguous()
    mask = torch.empty((h, w), device=yh.device, dtype=torch.uint8)
    check(_lib.lib().wmd_mask_threshold(ptr(yh), ptr(mm), float(thresh_ratio), ptr(mask), h, w, current_stream()),
          "wmd_mask_threshold")
    return mask


def _spec_array(specs, outs, counts):
    """counts = (int32 tensor [B,k] (or [k]), zero-initialised by the caller; [spec index of column 0, of column 1, ...]): the
    number of set pixels of those specs' masks is ADDED to the tensor by the same launch (wmd_dilate_spec.nnz)."""
    col = {}
    if counts is not None:
        t, which = counts
        if t.dtype != torch.int32 or not t.is_contiguous() or t.shape[-1] != len(which):
            raise _lib.WmdError("mask counts: a contiguous int32 tensor with one column per counted spec")
        col = {si: j for j, si in enumerate(which)}
    return (_lib.DilateSpec * len(specs))(*[
        _lib.DilateSpec(up, r, ptr(o), (counts[0].data_ptr() + 4 * col[i]) if i in col else None, len(col))
        for i, ((up, r), o) in enumerate(zip(specs, outs))])


def dilate_multi(mask, specs, counts=None):
    """mask uint8 [h,w] (or [B,h,w]: one mask per frame); specs = [(up, radius), ...] -> list of uint8 masks
    [h*up, w*up] (or [B,h*up,w*up]), one launch.  counts: see _spec_array."""
    _on_gpu(mask)
    batched = mask.dim() == 3
    B = mask.shape[0] if batched else 1
    h, w = mask.shape[-2:]
    mask = mask.contiguous()
    outs = [torch.empty(((B,) if batched else ()) + (h * up, w * up), device=mask.device, dtype=torch.uint8) for up, _ in specs]
    arr = _spec_array(specs, outs, counts)
    check(_lib.lib().wmd_mask_dilate_multi_b(ptr(mask), B, h, w, arr, len(specs), current_stream()), "wmd_mask_dilate_multi")
    return outs


def mask_level(yl, yh, thresh_ratio, specs, counts=None):
    """minmax(yl) -> threshold(yh) -> every dilated variant, one launch (depth_decoder.py:308-319).
    specs = [(up, radius), ...]; (1, 0) is the thresholded mask itself.  Bit-identical to the three separate calls.
    yh [B,1,3,h,w], yl [B,1,*,*]: B > 1 = one range / threshold / mask set per frame, outputs [B,h*up,w*up].
    counts: see _spec_array."""
    _on_gpu(yl, yh)
    h, w = yh.shape[-2:]
    B = yh.shape[0] if yh.dim() == 5 else 1
    yl, yh = yl.contiguous(), yh.contiguous()
    outs = [torch.empty(((B,) if B > 1 else ()) + (h * up, w * up), device=yh.device, dtype=torch.uint8) for up, _ in specs]
    arr = _spec_array(specs, outs, counts)
    mm = torch.empty((B, 2), device=yh.device, dtype=torch.float32) if B > 1 else None
    check(_lib.lib().wmd_mask_level_b(ptr(yl), yl.numel() // B, ptr(yh), float(thresh_ratio), B, h, w, arr, len(specs), ptr(mm),
                                      current_stream()), "wmd_mask_level")
    return outs


class LevelState:
    """Device-side state of the work-list form of the sparse levels (wmd_mask_level_lists), private to one decoder and one
    input signature, used in stream order:
      scratch  accumulators + ticket + forward counter of the mask launches (all zero at rest, never refilled),
      ring     RING slots of pixel counts: forward k's counts are published to slot k % RING by the mask launches' last
               workgroups and stamped k + 1 -- fetched only if somebody reads `total_ops` (no fill, no copy, no sync in a forward),
      keys     order-preserving (min, max) keys of every frame's low-pass plane, maintained by the head epilogues,
      pool     the activation planes of the sparse levels: zero-filled ONCE.  A tile the block-sparse kernels skip keeps the
               (finite) values of an earlier forward; nothing reads them -- every consumer masks its reads or selects on
               its mask (wmd_conv_args.in_mask, wmd_head_shiftsum_args.yh_mask)."""
    RING = 64

    def __init__(self, dev, B, n_levels, pool_floats):
        self.B, self.n_levels = B, n_levels
        self.scratch = torch.zeros(int(_lib.lib().wmd_mask_level_scratch_ints(B)), device=dev, dtype=torch.int32)
        self.slot_ints = 1 + max(n_levels, 1) * B * 3
        self.ring = torch.zeros(self.RING * self.slot_ints, device=dev, dtype=torch.int32)
        self.keys = torch.tensor([[-1, 0]] * B, device=dev, dtype=torch.int32)      # 0xFFFFFFFF, 0: the armed state
        self.pool = torch.zeros(max(int(pool_floats), 1), device=dev, dtype=torch.float32)
        self.lists = {}
        self.seq = 0               # host mirror of the device's forward counter
        self.pending = {}          # ring slot -> weak reference to the fetch of the forward that used it last

    def tile_list(self, key, capacity):
        hit = self.lists.get(key)
        if hit is None or hit[0].numel() < capacity:
            dev = self.scratch.device
            hit = self.lists[key] = (torch.zeros(max(int(capacity), 1), device=dev, dtype=torch.int32),
                                     torch.zeros(self.B, device=dev, dtype=torch.int32))
        return hit

    def counts_fetcher(self, k):
        """-> callable returning forward k's [n_levels][B*3] counts (python ints); to be called after forward k was
        enqueued.  The slot is read when the counts are wanted; a forward RING later would overwrite it, so the fetch of
        forward k is forced (cheaply: that forward is long complete) when forward k + RING is about to start."""
        import weakref
        ev = torch.cuda.Event()
        ev.record()
        slot = k % self.RING
        box = {}
        ring, slot_ints, n_levels, B = self.ring, self.slot_ints, self.n_levels, self.B

        def fetch():
            if "v" not in box:
                ev.synchronize()
                vals = ring[slot * slot_ints:(slot + 1) * slot_ints].cpu().tolist()
                if vals[0] != k + 1:
                    raise _lib.WmdError("sparse decoder: the pixel counts of forward %d are gone (ring slot stamped %d): "
                                        "read total_ops within %d forwards" % (k, vals[0] - 1, LevelState.RING))
                box["v"] = [[int(v) for v in vals[1 + l * B * 3:1 + (l + 1) * B * 3]] for l in range(n_levels)]
            return box["v"]
        self.pending[slot] = weakref.ref(fetch)
        return fetch

    def before_forward(self):
        """Called when forward number self.seq is about to be enqueued: the slot it will publish to must have been read."""
        for ahead in range(3):       # (a graph capture executes two warm-up forwards before the replay)
            old = self.pending.pop((self.seq + ahead) % self.RING, None)
            fetch = old() if old is not None else None
            if fetch is not None:
                fetch()


def mask_level_lists(state, B, h, w, specs, thresh_ratio=0.0, yl=None, yh=None, mask0=None, use_keys=False, counts_off=0,
                     advance=False):
    """One launch: threshold (or injected base mask [B,h,w]) -> every dilated variant + the active-tile work lists + the
    pixel counts, see wmd_mask_level_lists in include/wmd.h.  specs = [(up, radius, count column or 0, (tile_h, tile_w) or
    None[, and_mask uint8 [B,h*up,w*up]]), ...] -> (masks [B,h*up,w*up] uint8, lists [(list, count, tile_h, tile_w) or None])."""
    _on_gpu(yl, yh, mask0)
    dev = state.scratch.device
    outs, lists, arr = [], [], []
    ncounts = 0
    for j, spec in enumerate(specs):
        up, r, col, tile = spec[:4]
        andm = spec[4] if len(spec) > 4 else None
        if andm is not None and (andm.dtype != torch.uint8 or andm.numel() != B * h * up * w * up or not andm.is_contiguous()):
            raise _lib.WmdError("mask_level_lists: and_mask must be a contiguous uint8 [B,%d,%d] tensor" % (h * up, w * up))
        o = torch.empty((B, h * up, w * up), device=dev, dtype=torch.uint8)
        outs.append(o)
        ncounts = max(ncounts, col)
        if tile is not None:
            th, tw = tile
            tl, tc = state.tile_list((h * up, w * up, th, tw, j), B * (-(-h * up // th)) * (-(-w * up // tw)))
            lists.append((tl, tc, th, tw))
            arr.append(_lib.LevelSpec(up, r, ptr(o), col, th, tw, ptr(tl), ptr(tc), ptr(andm)))
        else:
            lists.append(None)
            arr.append(_lib.LevelSpec(up, r, ptr(o), col, 0, 0, None, None, ptr(andm)))
    if mask0 is not None:
        mask0 = mask0.contiguous()
    else:
        yl, yh = yl.contiguous(), yh.contiguous()
    a = _lib.MaskLevelArgs(B=B, h=h, w=w, yl=ptr(yl), n_yl=(yl.numel() // B) if yl is not None else 0, yh=ptr(yh),
                           thresh_ratio=float(thresh_ratio), mask0=ptr(mask0), minmax=None,
                           range_keys=ptr(state.keys) if use_keys else None,
                           specs=(_lib.LevelSpec * len(arr))(*arr), n=len(arr), scratch=ptr(state.scratch), counts=ptr(state.ring),
                           ring_slots=state.RING, slot_ints=state.slot_ints, counts_off=int(counts_off), ncounts=ncounts,
                           advance=int(bool(advance)))
    check(_lib.lib().wmd_mask_level_lists(C.byref(a), current_stream()), "wmd_mask_level_lists")
    return outs, lists


def compact_multi(masks, nnz_out=None):
    """uint8 masks [h,w] (or [B,h,w]) -> (list of int32 coordinate lists [npix capacity] (or [B,npix]), int32 tensor of
    counts [n] (or [B,n])) in one launch; raster order, counts stay on the device.  nnz_out: a contiguous int32 tensor of that
    shape to receive the counts (callers that collect the counts of several calls in one buffer)."""
    _on_gpu(*masks)
    n = len(masks)
    batched = masks[0].dim() == 3
    B = masks[0].shape[0] if batched else 1
    masks = [m.contiguous() for m in masks]
    nnz = nnz_out if nnz_out is not None else torch.empty((B, n) if batched else (n,), device=masks[0].device, dtype=torch.int32)
    if nnz.dtype != torch.int32 or not nnz.is_contiguous() or nnz.numel() != B * n:
        raise _lib.WmdError("compact_multi: nnz_out must be a contiguous int32 tensor of %d elements" % (B * n))
    coords = [torch.empty((B, m.numel() // B) if batched else (m.numel(),), device=m.device, dtype=torch.int32) for m in masks]
    arr = (_lib.CompactSpec * n)(*[_lib.CompactSpec(ptr(m), m.numel() // B, ptr(c), nnz.data_ptr() + 4 * i)
                                   for i, (m, c) in enumerate(zip(masks, coords))])
    check(_lib.lib().wmd_mask_compact_multi_b(arr, n, B, current_stream()), "wmd_mask_compact_multi")
    return coords, nnz


def sparse_conv(y, x1, wp, bias, cout, ksize, out_coords, out_nnz, max_out, x2=None, up1=1, in_mask=None,
                pad="reflect", act="none", slope=0.0, out_scale=1.0, c1=None, c1_off=0, wp2=None, bias2=None,
                c1_off2=0, split_waves=0, nnz_stride=0):
    """Gather-GEMM convolution on the active pixels; writes y [Cout,H,W] in place at those pixels.
    Batched form: y [B,Cout,H,W], x1 [B,C,h,w], x2 [B,C2,H,W], in_mask [B,H,W], out_coords [B,max_out]; out_nnz is the
    address of frame 0's count and nnz_stride the number of int32 between the counts of consecutive frames."""
    _on_gpu(y, x1, x2, wp, bias, in_mask, out_coords, wp2, bias2)
    B = y.shape[0] if y.dim() == 4 else 1
    Cout_, H, W = y.shape[-3:]
    assert Cout_ == cout
    c1tot = x1.shape[-3]
    c1 = c1tot if c1 is None else c1
    a = _lib.SparseConvArgs(H=H, W=W, C1=c1, up1=up1, C1tot=c1tot, c1_off=c1_off, C2=0 if x2 is None else x2.shape[-3],
                            Cout=cout, ksize=ksize, pad_mode=PAD[pad], act=ACT[act], slope=float(slope),
                            x1=ptr(x1), x2=ptr(x2), in_mask=ptr(in_mask), out_coords=ptr(out_coords),
                            out_nnz=out_nnz, max_out=int(max_out), wp=ptr(wp), bias=ptr(bias), wp2=ptr(wp2),
                            bias2=ptr(bias2), c1_off2=c1_off2, out_scale=float(out_scale), y=ptr(y),
                            split_waves=int(split_waves), B=B, nnz_stride=int(nnz_stride))
    check(_lib.lib().wmd_sparse_conv(C.byref(a), current_stream()), "wmd_sparse_conv")
    return y


class LazyOpsDict(dict):
    """The output dictionary of the sparse decoders.  The reference's `total_ops` entries are python ints computed from
    pixel counts that live on the device; resolving them costs a host synchronisation, which the forward itself no longer
    pays: the entries are computed on first access (`out["total_ops"]`, `.items()`, `.values()`, `.get`, `==` ...), from
    counts copied to pinned host memory behind an event recorded by the forward.  Consumers that only read the maps never
    stall the stream; consumers that read the op counts see exactly the integers the reference returns."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._lazy = None      # callable -> {key: int}

    def set_lazy(self, keys, resolver):
        for key in keys:
            dict.__setitem__(self, key, None)
        self._lazy = resolver

    def resolve(self):
        if self._lazy is not None:
            fn, self._lazy = self._lazy, None
            for key, v in fn().items():
                dict.__setitem__(self, key, v)
        return self

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if v is None and self._lazy is not None:
            self.resolve()
            v = dict.__getitem__(self, key)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return dict.items(self.resolve())

    def values(self):
        return dict.values(self.resolve())

    def copy(self):
        return dict(self.resolve())

    # CPython copies a dict subclass through the raw hash table -- `dict(out)`, `{**out}`, `out | x`, dict.update(x, out) --
    # unless the subclass overrides __iter__: then it goes through keys() + __getitem__, i.e. the lazy VALUES resolve on access
    # while plain key iteration (`for k in out`, `list(out)`) never synchronises.  The remaining readers of the raw storage
    # resolve first.
    def __iter__(self):         # keys are known up front: iterating them costs no synchronisation (the values resolve in __getitem__)
        return dict.__iter__(self)

    def keys(self):
        return dict.keys(self)

    def pop(self, *a):
        return dict.pop(self.resolve(), *a)

    def popitem(self):
        return dict.popitem(self.resolve())

    def setdefault(self, key, default=None):
        return dict.setdefault(self.resolve(), key, default)

    def __repr__(self):
        return dict.__repr__(self.resolve())

    def __reduce__(self):      # pickle / copy.copy / copy.deepcopy: a plain dict with the integers filled in
        return (dict, (dict(self.resolve()),))

    def __eq__(self, other):
        return dict.__eq__(self.resolve(), other.resolve() if isinstance(other, LazyOpsDict) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


class _PinnedSlot:
    __slots__ = ("buf", "ev")


_pinned_free = []


def _pinned_acquire(n, dtype):
    """A pinned host buffer of >= n elements from a small pool.  A fresh `torch.empty(pin_memory=True)` per forward goes through
    the caching host allocator, which only recycles a block whose last copy has completed -- a burst of asynchronous decodes
    outruns it and every now and then pays a hipHostMalloc of tens of milliseconds inside somebody's timed loop."""
    for i, sl in enumerate(_pinned_free):
        if sl.buf.dtype == dtype and sl.buf.numel() >= n:
            del _pinned_free[i]
            sl.ev.synchronize()       # (long complete: the slot was released by its reader)
            return sl
    sl = _PinnedSlot()
    sl.buf = torch.empty(max(int(n), 256), dtype=dtype, pin_memory=True)
    sl.ev = torch.cuda.Event()
    return sl


def _pinned_release(slot):
    if len(_pinned_free) < 64:
        _pinned_free.append(slot)


def counts_to_host(count_tensors):
    """Start an asynchronous copy of small device count tensors into pinned host memory; -> callable returning them as
    lists of python ints (waits for the copy only)."""
    import torch
    if not count_tensors:
        return lambda: []
    first = count_tensors[0]
    expect, adjacent = first.data_ptr(), True
    for c in count_tensors:          # views that tile one buffer in order: copy the buffer, no gather launch
        adjacent = adjacent and c.is_contiguous() and c.data_ptr() == expect and c.dtype == first.dtype
        expect += c.numel() * c.element_size()
    if adjacent and len(count_tensors) > 1:
        flat = torch.as_strided(first, (sum(c.numel() for c in count_tensors),), (1,), first.storage_offset())
    else:
        flat = torch.cat([c.reshape(-1) for c in count_tensors])
    slot = _pinned_acquire(flat.numel(), flat.dtype)
    host = slot.buf[:flat.numel()]
    host.copy_(flat, non_blocking=True)
    slot.ev.record()
    sizes = [c.numel() for c in count_tensors]

    def fetch():
        slot.ev.synchronize()
        vals, out, o = host.tolist(), [], 0
        for n in sizes:
            out.append([int(v) for v in vals[o:o + n]])
            o += n
        return out
    import weakref
    weakref.finalize(fetch, _pinned_release, slot)      # the slot returns to the pool when the last user of `fetch` is gone
    return fetch
