"""Operator layer: torch tensors in, torch tensors out, every FLOP in libwmd_hip.so.

Each function mirrors one reference operator group (file:line under /root/reference):
  conv2d_fused   ConvBlock / Conv3x3 / Conv1x1 + upsample + skip concat
                 (KITTI/layers.py:120-173,233-236; depth_decoder.py:145-150;
                  NYUv2/networks/layers.py:11-32,57-67)
  head3x3        last Conv3x3 of the wavelet heads + sigmoid + 2^(s-1)(s+ - s-)
                 (depth_decoder.py:104-136; densedepth_decoder.py:106-115)
  idwt_haar      IDWT(wave="haar", mode="zero") + disparity normalisation (depth_decoder.py:164-166)
  dwt_haar       DWT(J, "haar", "reflect") on even sizes (NYUv2/train.py:258,289)
All of them are differentiable (torch.autograd.Function with hand-written HIP backward kernels).
"""
import ctypes as C
import os

import torch

from . import _lib, tuner
from ._lib import ACT, PAD, check, current_stream, ptr


def _require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.WmdError("wavelet_monodepth_amd ops run on the MI355X only (got a %s tensor); "
                                "there is no CPU implementation in the package" % t.device)
        if t.dtype != torch.float32:
            raise _lib.WmdError("fp32 tensors expected, got %s" % t.dtype)


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------
# weight packing
# ---------------------------------------------------------------------------------------------

_PACK_CACHE = os.environ.get("WMD_PACK_CACHE", "1") != "0"
_pack_generation = [0]


def pack_generation():
    return _pack_generation[0]


def invalidate_packs():
    """Forget every memoised packed-weight image and every captured hipGraph keyed on weights.

    The memo tags are (autograd version counter, data_ptr): optimizer steps, `load_state_dict`, `copy_` & co. bump the
    counter and are picked up automatically.  Writes that go through `.data` (`p.data.mul_()` for EMA, manual init) or
    through raw pointers do NOT bump it — call this afterwards (or update the weights with version-bumping ops), otherwise
    the forward keeps multiplying by the stale image."""
    _pack_generation[0] += 1


def pack_weights(weight, dgrad=False):
    """[Cout,Cin,k,k] -> MFMA fragment image (wmd_conv_pack_weights[_dgrad]).

    The image is memoised ON the weight tensor object, keyed by its autograd version counter and storage
    pointer, so it is rebuilt whenever the weight is updated in place (optimizer step, load_state_dict) and
    reused while it is constant (inference).  WMD_PACK_CACHE=0 disables the memo."""
    l = _lib.lib()
    tag = (weight._version, weight.data_ptr(), weight.device, _pack_generation[0])
    slot = "_wmd_pack_d" if dgrad else "_wmd_pack_f"
    if _PACK_CACHE:
        hit = getattr(weight, slot, None)
        if hit is not None and hit[0] == tag:
            return hit[1]
    cout, cin, k, _ = weight.shape
    n = l.wmd_conv_packed_weight_floats(cout, cin, k)
    wp = torch.empty(n, device=weight.device, dtype=torch.float32)
    fn = l.wmd_conv_pack_weights_dgrad if dgrad else l.wmd_conv_pack_weights
    check(fn(ptr(_c(weight.detach())), ptr(wp), cout, cin, k, current_stream()), "wmd_conv_pack_weights")
    if _PACK_CACHE and not torch.cuda.is_current_stream_capturing():
        try:
            setattr(weight, slot, (tag, wp))
        except AttributeError:
            pass
    return wp


_WINOGRAD = os.environ.get("WMD_WINOGRAD", "1") != "0"
_PREPACK = os.environ.get("WMD_PREPACK", "1") != "0"


def pack_many(specs, with_buffer=False):
    """specs: [(weight [Cout,Cin,k,k], ("fwd" | "dgrad" | "wino_fwd" | "wino_dgrad", ...)), ...] -> one dict {image name:
    tensor} per spec, all written by ONE wmd_conv_pack_many launch into one buffer, in the order given (no memo).
    with_buffer: -> (dicts, that buffer)."""
    l = _lib.lib()
    sizes = []
    for w, names in specs:
        cout, cin, k, _k = w.shape
        n_direct, n_wino = l.wmd_conv_packed_weight_floats(cout, cin, k), l.wmd_conv_packed_weight_floats_wino(cout, cin)
        sizes.append([n_wino if nm.startswith("wino") else n_direct for nm in names])
    buf = torch.empty(sum(sum(s) for s in sizes), device=specs[0][0].device, dtype=torch.float32)
    items, out, keep, off = [], [], [], 0
    for (w, names), sz in zip(specs, sizes):
        cout, cin, k, _k = w.shape
        wc = _c(w.detach())
        keep.append(wc)
        it = _lib.PackItem(w=ptr(wc), Cout=cout, Cin=cin, ksize=k)
        d = {}
        for nm, n in zip(names, sz):
            d[nm] = buf[off:off + n]
            setattr(it, nm, d[nm].data_ptr())
            off += n
        items.append(it)
        out.append(d)
    arr = (_lib.PackItem * len(items))(*items)
    check(l.wmd_conv_pack_many(arr, len(items), current_stream()), "wmd_conv_pack_many")
    return (out, buf) if with_buffer else out


def prepack(weights, dgrad=True):
    """Build every weight image the convolutions of `weights` ([Cout,Cin,k,k] tensors) will ask for -- forward and (dgrad)
    data-gradient fragment order, direct and Winograd -- in ONE launch (wmd_conv_pack_many) into one buffer, and memoise
    them on the tensors exactly as pack_weights / pack_weights_wino would.  A training step calls this once per forward
    (the optimizer step invalidated all ~50 images); weights whose images are current are skipped, so inference pays
    nothing.  WMD_PREPACK=0: every convolution packs for itself as before."""
    if not (_PREPACK and _PACK_CACHE) or not weights or torch.cuda.is_current_stream_capturing():
        return
    todo, seen = [], set()
    for w in weights:
        if w is None or id(w) in seen or not w.is_cuda or w.dim() != 4 or w.shape[2] not in (1, 3):
            continue
        seen.add(id(w))
        tag = (w._version, w.data_ptr(), w.device, _pack_generation[0])
        k = w.shape[2]
        slots = ["_wmd_pack_f"] + (["_wmd_pack_d"] if dgrad else [])
        if k == 3 and _WINOGRAD:
            slots += ["_wmd_pack_wf"] + (["_wmd_pack_wd"] if dgrad else [])
        need = [sl for sl in slots if getattr(w, sl, (None,))[0] != tag]
        if need:
            todo.append((w, tag, need))
    if not todo:
        return
    field = {"_wmd_pack_f": "fwd", "_wmd_pack_d": "dgrad", "_wmd_pack_wf": "wino_fwd", "_wmd_pack_wd": "wino_dgrad"}
    images = pack_many([(w, tuple(field[sl] for sl in need)) for w, _, need in todo])
    views = [(w, sl, tag, img[field[sl]]) for (w, tag, need), img in zip(todo, images) for sl in need]
    for w, sl, tag, view in views:
        try:
            setattr(w, sl, (tag, view))
        except AttributeError:
            pass




def prepack_module(module):
    """prepack() for every ungrouped convolution filter below `module` (the decoders call this first thing in forward)."""
    if not (_PREPACK and _PACK_CACHE):
        return
    ws = getattr(module, "_wmd_prepack_list", None)
    if ws is None:
        # (the 1- and 3-channel 3x3 output layers run on the head kernels, which read the raw filter)
        ws = [m.weight for m in module.modules() if isinstance(m, torch.nn.Conv2d) and m.groups == 1
              and not (m.kernel_size[0] == 3 and m.out_channels <= 4)]
        module.__dict__["_wmd_prepack_list"] = ws      # plain attribute: parameters are replaced only by re-construction
    if ws and all(w is not None and w.is_cuda for w in ws):      # (CPU tensors: the first operator refuses them)
        prepack(ws, dgrad=torch.is_grad_enabled())


def pack_weights_wino(weight, dgrad=False):
    """[Cout,Cin,3,3] -> Winograd F(2x2,3x3) fragment image U = G g G^T (wmd_conv_pack_weights_wino); memoised like
    pack_weights.  Returns None for other kernel sizes or when WMD_WINOGRAD=0."""
    if not _WINOGRAD or weight.shape[-1] != 3:
        return None
    l = _lib.lib()
    tag = (weight._version, weight.data_ptr(), weight.device, _pack_generation[0])
    slot = "_wmd_pack_wd" if dgrad else "_wmd_pack_wf"
    if _PACK_CACHE:
        hit = getattr(weight, slot, None)
        if hit is not None and hit[0] == tag:
            return hit[1]
    cout, cin = weight.shape[:2]
    rows, red = (cin, cout) if dgrad else (cout, cin)
    wp = torch.empty(l.wmd_conv_packed_weight_floats_wino(rows, red), device=weight.device, dtype=torch.float32)
    check(l.wmd_conv_pack_weights_wino(ptr(_c(weight.detach())), ptr(wp), cout, cin, int(dgrad), current_stream()),
          "wmd_conv_pack_weights_wino")
    if _PACK_CACHE and not torch.cuda.is_current_stream_capturing():
        try:
            setattr(weight, slot, (tag, wp))
        except AttributeError:
            pass
    return wp


# ---------------------------------------------------------------------------------------------
# dense convolution
# ---------------------------------------------------------------------------------------------

def _conv_fwd_raw(x1, x2, wp, bias, cout, ksize, pad, act, slope, up1, wp_wino=None, in_mask=None, out_mask=None, out=None,
                  in_mask_2x2=False, out_tiles=None):
    """in_mask / out_mask (uint8 [B,H,W]) + out (zero-initialised [B,cout,H,W]): block-sparse execution, see
    wmd_conv_args.in_mask in include/wmd.h; in_mask_2x2: the caller's promise that in_mask is constant on 2x2 blocks.
    out_tiles = (list int32, count int32 scalar, tile_h, tile_w): the work-list form (wmd_conv_args.out_tiles, built by
    sparse_ops.mask_level_lists) -- only listed tiles are computed, the split of the reduction is chosen on the device.
    """
    l = _lib.lib()
    B, C1 = x1.shape[0], x1.shape[1]
    H, W = x1.shape[2] * up1, x1.shape[3] * up1
    C2 = 0 if x2 is None else x2.shape[1]
    if x2 is not None and (x2.shape[0] != B or x2.shape[2] != H or x2.shape[3] != W):
        raise _lib.WmdError("skip tensor %s does not match upsampled input %s" % (tuple(x2.shape), (B, C1, H, W)))
    if out_mask is not None and out is None:
        raise _lib.WmdError("block-sparse convolution writes the active tiles only: pass a zero-initialised `out`")
    y = out if out is not None else torch.empty((B, cout, H, W), device=x1.device, dtype=torch.float32)
    a = _lib.ConvArgs(B=B, H=H, W=W, C1=C1, up1=up1, C2=C2, Cout=cout, ksize=ksize, pad_mode=PAD[pad], act=ACT[act],
                      slope=float(slope), x1=ptr(x1), x2=ptr(x2), wp=ptr(wp), bias=ptr(bias), y=ptr(y),
                      workspace=None, workspace_floats=0, tune_cfg=0, tune_ksplit=0, wp_wino=ptr(wp_wino),
                      in_mask=ptr(in_mask), out_mask=ptr(out_mask), in_mask_2x2=int(bool(in_mask_2x2)))
    if out_tiles is not None:
        tl, tc, th, tw = out_tiles
        a.out_tiles, a.out_tile_count, a.out_tile_h, a.out_tile_w = ptr(tl), ptr(tc), int(th), int(tw)
    stream = current_stream()
    keep = []

    def launch(cfg, ks):
        a.tune_cfg, a.tune_ksplit = cfg, ks
        a.workspace, a.workspace_floats = None, 0
        n = l.wmd_conv_fwd_workspace_floats(C.byref(a))
        if n:
            ws = torch.empty(n, device=x1.device, dtype=torch.float32)
            keep[:] = [ws]   # same-stream reuse: the caching allocator may hand the block to the next candidate
            a.workspace, a.workspace_floats = ptr(ws), n
        elif ks > 1:
            return -3
        return l.wmd_conv_fwd(C.byref(a), stream)

    choice = (0, 0)
    if tuner.enabled and out_tiles is None:     # (a work list fixes the tile shape; the K split is the device's decision)
        key = "conv|%d|%d|%d|%d|%d|%d|%d|%d" % (B, H, W, C1, up1, C2, cout, ksize)  # str: JSON-cacheable
        if wp_wino is None and ksize == 3:
            key += "|direct"   # a choice made with the Winograd configurations on offer must not be reused without them
        if out_mask is not None:
            key += "|tiles"    # block-sparse execution: its own (masked) instantiations, timed at full density (below)
        choice = tuner.lookup(key)
        if choice is None:
            if torch.cuda.is_current_stream_capturing():
                choice = (0, 0)  # cannot time inside a capture: the library's cost model decides
            elif out_mask is not None or in_mask is not None:
                # The candidates are timed on ALL-ONES masks into a scratch output, not on the caller's masks: a first call with
                # empty masks (every block returns at once, every configuration "takes" the same few microseconds) used to
                # fix an arbitrary choice for the life of the process -- seen as 41 + 62 us unsplit level-2 launches after a
                # test had tuned the same shape on an empty mask.  Full density is the reproducible worst case.
                saved = (a.in_mask, a.out_mask, a.y)
                ones = torch.ones((B, H, W), device=x1.device, dtype=torch.uint8)
                scratch = torch.empty_like(y)
                a.in_mask = ptr(ones) if in_mask is not None else None
                a.out_mask = ptr(ones) if out_mask is not None else None
                a.y = ptr(scratch)
                try:
                    choice = tuner.tune(key, 9 if ksize == 3 else 1, launch)
                finally:
                    a.in_mask, a.out_mask, a.y = saved
            else:
                choice = tuner.tune(key, 9 if ksize == 3 else 1, launch)
    check(launch(*choice), "wmd_conv_fwd")
    return y


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, weight, bias, ksize, pad, act, slope, up1, x1_gate=None, grad_is_dz=False):
        x1c, x2c = _c(x1), _c(x2)
        wp = pack_weights(weight)
        y = _conv_fwd_raw(x1c, x2c, wp, _c(bias), weight.shape[0], ksize, pad, act, slope, up1, pack_weights_wino(weight))
        ctx.save_for_backward(x1c, x2c, weight, y)
        ctx.has_x2 = x2 is not None
        ctx.has_bias = bias is not None
        ctx.cfg = (ksize, pad, act, slope, up1)
        ctx.x1_gate, ctx.grad_is_dz = x1_gate, grad_is_dz
        return y

    @staticmethod
    def backward(ctx, dy):
        l = _lib.lib()
        x1, x2, weight, y = ctx.saved_tensors
        ksize, pad, act, slope, up1 = ctx.cfg
        s = current_stream()
        dy = _c(dy)
        B, cout, H, W = y.shape
        C1 = x1.shape[1]
        C2 = 0 if x2 is None else x2.shape[1]
        if ACT[act] != 0 and not ctx.grad_is_dz:
            dz = torch.empty_like(dy)
            check(l.wmd_act_bwd(ptr(dy), ptr(y), ptr(dz), dy.numel(), ACT[act], float(slope), s), "wmd_act_bwd")
        else:
            dz = dy     # no activation, or every consumer already multiplied its contribution by f'(y) (x1_gate on their side)
        gate_act, gate_slope = (ACT[ctx.x1_gate[0]], float(ctx.x1_gate[1])) if ctx.x1_gate else (0, 0.0)
        need_x1, need_x2, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and x2 is not None, ctx.needs_input_grad[2]
        dx1 = dx2 = dw = db = None
        if need_x1 or need_x2:
            wpd = pack_weights(weight, dgrad=True)
            wpdw = pack_weights_wino(weight, dgrad=True)   # kept alive in this scope for the launch
            dx1 = torch.empty_like(x1) if need_x1 else None
            dx2 = torch.empty_like(x2) if need_x2 else None
            a = _lib.ConvDgradArgs(B=B, H=H, W=W, C1=C1, up1=up1, C2=C2, Cout=cout, ksize=ksize, pad_mode=PAD[pad],
                                   dz=ptr(dz), wp_dgrad=ptr(wpd), dx1=ptr(dx1), dx2=ptr(dx2), workspace=None,
                                   workspace_floats=0, tune_cfg=0, tune_ksplit=0, wp_dgrad_wino=ptr(wpdw),
                                   x1_fwd=ptr(x1) if gate_act else None, x1_act=gate_act, x1_slope=gate_slope)
            _dgrad_launch(a, dy.device, 9 if ksize == 3 else 1)
        if need_w or (ctx.has_bias and ctx.needs_input_grad[3]):
            dw = torch.empty_like(weight)
            db = torch.empty(cout, device=dy.device, dtype=torch.float32) if ctx.has_bias else None
            a = _lib.ConvWgradArgs(B=B, H=H, W=W, C1=C1, up1=up1, C2=C2, Cout=cout, ksize=ksize, pad_mode=PAD[pad],
                                   x1=ptr(x1), x2=ptr(x2), dz=ptr(dz), dw=ptr(dw), dbias=ptr(db), workspace=None,
                                   workspace_floats=0, tune_cfg=0, tune_nsplit=0)
            _wgrad_launch(a, dy.device)
        return dx1, dx2, dw, db, None, None, None, None, None, None, None


_WGRAD_NAMES = None


def _wgrad_launch(a, device):
    """wmd_conv_wgrad with the kernel family / tile chosen per problem signature on the device (3x3: the Winograd
    F(2x2,3x3) weight-gradient tiles against the direct kernel), like the forward and the data gradient."""
    global _WGRAD_NAMES
    l = _lib.lib()
    stream = current_stream()
    keep = []

    def launch(cfg):
        a.tune_cfg, a.tune_nsplit = cfg, 0
        n = l.wmd_conv_wgrad_workspace_floats(C.byref(a))
        ws = torch.empty(max(n, 1), device=device, dtype=torch.float32)
        keep[:] = [ws]
        a.workspace, a.workspace_floats = ptr(ws), n
        return l.wmd_conv_wgrad(C.byref(a), stream)

    cfg = 0
    if tuner.enabled and a.ksize == 3:
        if _WGRAD_NAMES is None:
            _WGRAD_NAMES = [l.wmd_conv_wgrad_config_name(i).decode() for i in range(l.wmd_conv_wgrad_num_configs())]
        key = "wgrad|%d|%d|%d|%d|%d|%d|%d|%d" % (a.B, a.H, a.W, a.C1, a.up1, a.C2, a.Cout, a.ksize)
        cands = [("library", 0), ("direct", -1)] + [(n, i + 1) for i, n in enumerate(_WGRAD_NAMES)]
        cfg = tuner.choose(key, cands, launch)
    check(launch(cfg), "wmd_conv_wgrad")


def conv2d_pre_activated(x1, x1_pre, weight, bias=None, up1=1, pad="reflect", act="none", slope=0.0):
    """conv2d_fused for an x1 that is still a PRE-activation (the encoder edge, layers.DeferredActivation):
    act( conv( pad( nearest_up( pre(x1) ) ) ) + bias ) with pre(v)[c] = pre_act(v * scale[c] + shift[c]); x1_pre = (scale, shift,
    act, slope).  Round 3 applied pre() on load inside dedicated instantiations of the direct kernel; that form lost the tuned
    Winograd / split-K choice and measured SLOWER than one elementwise pass + the tuned convolution (tools/probes/edge_microbench.py:
    R18 640x192 b12 75.4 vs 66.9 us, R50 1024x320 b8 432.9 vs 170.5 us), so the edge is activated explicitly here and the
    instantiations are gone (round 4).  Inference operator, like the decoders' use of it."""
    _require_gpu(x1, weight, bias)
    if torch.is_grad_enabled() and (x1.requires_grad or weight.requires_grad):
        raise _lib.WmdError("conv2d_pre_activated is an inference operator (activate the tensor and use conv2d_fused to train)")
    if weight.shape[1] != x1.shape[1]:
        raise _lib.WmdError("weight expects %d input channels, got %d" % (weight.shape[1], x1.shape[1]))
    psc, psh, pact, pslope = x1_pre
    _require_gpu(psc, psh)
    if pact not in ("none", "leaky", None):
        raise _lib.WmdError("conv2d_pre_activated: the edge's activation is 'none' or 'leaky' (ReLU = slope 0), got %r" % (pact,))
    for v in (psc, psh):
        if v is not None and (v.dtype != torch.float32 or v.numel() != x1.shape[1]):
            raise _lib.WmdError("x1_pre: scale / shift must be float32 tensors with one value per x1 channel")
    with torch.no_grad():
        v = x1
        if psc is not None:
            v = v * psc.view(1, -1, 1, 1)
        if psh is not None:
            v = v + psh.view(1, -1, 1, 1)
        if pact == "leaky":
            v = torch.relu(v) if pslope == 0.0 else torch.nn.functional.leaky_relu(v, pslope)
        return conv2d_fused(v, weight, bias, up1=up1, pad=pad, act=act, slope=slope)


def conv2d_fused(x1, weight, bias=None, x2=None, up1=1, pad="reflect", act="none", slope=0.0, x1_gate=None, grad_is_dz=False):
    """act( conv_kxk( pad( cat[ nearest_up(x1, up1), x2 ] ) ) + bias ), k taken from `weight`.

    Backward-only hints for callers that own the whole graph around this op (the decoders): `x1_gate=(act, slope)` says x1
    is the output of that activation and makes the data gradient return dx1 * act'(x1) -- the pre-activation gradient of
    x1's producer -- from the same kernel; `grad_is_dz=True` on that producer says EVERY consumer of its output does so,
    so its backward takes the incoming gradient as dz (no wmd_act_bwd pass).  The derivative is linear, so gating each
    consumer's contribution before autograd sums them equals gating the sum."""
    _require_gpu(x1, x2, weight, bias)
    ksize = weight.shape[-1]
    cin = x1.shape[1] + (0 if x2 is None else x2.shape[1])
    if weight.shape[1] != cin:
        raise _lib.WmdError("weight expects %d input channels, got %d" % (weight.shape[1], cin))
    return _ConvFn.apply(x1, x2, weight, bias, ksize, pad, act, slope, up1, x1_gate, grad_is_dz)


class _DwConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, weight, pad, up1):
        x1c, x2c, wc = _c(x1), _c(x2), _c(weight)
        B, C1 = x1c.shape[:2]
        H, W = x1c.shape[2] * up1, x1c.shape[3] * up1
        C2 = 0 if x2c is None else x2c.shape[1]
        y = torch.empty((B, C1 + C2, H, W), device=x1c.device, dtype=torch.float32)
        a = _lib.DwConvArgs(B=B, H=H, W=W, C1=C1, up1=up1, C2=C2, pad_mode=PAD[pad], x1=ptr(x1c), x2=ptr(x2c), w=ptr(wc))
        check(_lib.lib().wmd_dwconv3x3_fwd(C.byref(a), ptr(y), current_stream()), "wmd_dwconv3x3_fwd")
        ctx.save_for_backward(x1c, x2c, wc, y)
        ctx.cfg = (pad, up1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, w, y = ctx.saved_tensors
        pad, up1 = ctx.cfg
        l = _lib.lib()
        B, Cc, H, W = y.shape
        C1 = x1.shape[1]
        a = _lib.DwConvArgs(B=B, H=H, W=W, C1=C1, up1=up1, C2=Cc - C1, pad_mode=PAD[pad], x1=ptr(x1), x2=ptr(x2), w=ptr(w))
        dx1 = torch.empty_like(x1) if ctx.needs_input_grad[0] else None
        dx2 = torch.empty_like(x2) if (x2 is not None and ctx.needs_input_grad[1]) else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[2] else None
        n = l.wmd_dwconv3x3_bwd_workspace_floats(C.byref(a))
        ws = torch.empty(max(n, 1), device=y.device, dtype=torch.float32)
        check(l.wmd_dwconv3x3_bwd(C.byref(a), ptr(y), ptr(_c(dy)), ptr(dx1), ptr(dx2), ptr(dw), ptr(ws), n, current_stream()),
              "wmd_dwconv3x3_bwd")
        return dx1, dx2, dw, None, None


def dwconv3x3_relu(x1, weight, x2=None, up1=1, pad="zero"):
    """relu( depthwise_conv3x3( pad( cat[ nearest_up(x1, up1), x2 ] ) ) ), weight [C1+C2,1,3,3] -- the first half of the
    NYUv2 `is_depthwise` Conv3x3 (NYUv2/networks/layers.py:70-75).  Differentiable."""
    _require_gpu(x1, x2, weight)
    cin = x1.shape[1] + (0 if x2 is None else x2.shape[1])
    if tuple(weight.shape) != (cin, 1, 3, 3):
        raise _lib.WmdError("depthwise weight must be [%d,1,3,3], got %s" % (cin, tuple(weight.shape)))
    return _DwConvFn.apply(x1, x2, weight, pad, up1)


# ---------------------------------------------------------------------------------------------
# wavelet heads
# ---------------------------------------------------------------------------------------------

def _head_raw(xp, wp_, bp, xn, wn, bn, pad, mode, scale, save_sig):
    l = _lib.lib()
    B, Cc, H, W = xp.shape
    cout = wp_.shape[0]
    y = torch.empty((B, cout, H, W), device=xp.device, dtype=torch.float32)
    sp = torch.empty_like(y) if (save_sig and mode >= 1) else None
    sn = torch.empty_like(y) if (save_sig and mode == 2) else None
    a = _lib.HeadArgs(B=B, H=H, W=W, C=Cc, Cout=cout, pad_mode=PAD[pad], mode=mode, scale=float(scale),
                      xp=ptr(xp), wgt_p=ptr(wp_), bias_p=ptr(bp), xn=ptr(xn), wgt_n=ptr(wn), bias_n=ptr(bn),
                      y=ptr(y), sig_p=ptr(sp), sig_n=ptr(sn))
    _head_launch(l, a, xp.device)
    return y, sp, sn


def _head_launch(l, a, device):
    n = l.wmd_head3x3_workspace_floats(C.byref(a))
    if n:
        ws = torch.empty(n, device=device, dtype=torch.float32)
        a.workspace, a.workspace_floats = ptr(ws), n
    check(l.wmd_head3x3_fwd(C.byref(a), current_stream()), "wmd_head3x3_fwd")


class _HeadFn(torch.autograd.Function):
    """y = scale*conv(xp) | scale*sigmoid(conv(xp)) | scale*(sigmoid(conv_p(xp)) - sigmoid(conv_n(xn))).
    Backward reuses the generic MFMA dgrad/wgrad kernels on dz = dy*scale*sigma'(.)"""

    @staticmethod
    def forward(ctx, xp, wp_, bp, xn, wn, bn, pad, mode, scale, x_gate=None):
        ctx.x_gate = x_gate
        xp, xn = _c(xp), _c(xn)
        y, sp, sn = _head_raw(xp, _c(wp_), _c(bp), xn, _c(wn), _c(bn), pad, mode, scale,
                               save_sig=any(ctx.needs_input_grad))
        ctx.save_for_backward(xp, wp_, xn, wn, sp, sn)
        ctx.cfg = (pad, mode, scale, bp is not None, bn is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, wp_, xn, wn, sp, sn = ctx.saved_tensors
        pad, mode, scale, has_bp, has_bn = ctx.cfg
        dy = _c(dy)
        outs = [None] * 10
        sides = [(xp, wp_, sp, +1.0, 0, has_bp)]
        if mode == 2:
            sides.append((xn, wn, sn, -1.0, 3, has_bn))
        for x, w, sig, sign, base, has_b in sides:
            if mode == 0:
                dz = dy * (sign * scale)
            else:
                dz = dy * sig * (1.0 - sig) * (sign * scale)
            dx, dw, db = _conv_backward_raw(x, None, w, dz, 3, pad, 1, has_b,
                                            ctx.needs_input_grad[base], True, x1_gate=ctx.x_gate)
            outs[base], outs[base + 1], outs[base + 2] = dx, dw, db
        return tuple(outs)


def _dgrad_launch(a, device, taps):
    """wmd_conv_dgrad with the (tile, split-K) choice autotuned per problem signature, like the forward."""
    l = _lib.lib()
    stream = current_stream()
    keep = []

    def launch(cfg, ks):
        a.tune_cfg, a.tune_ksplit = cfg, ks
        n = l.wmd_conv_dgrad_workspace_floats(C.byref(a))
        ws = torch.empty(max(n, 1), device=device, dtype=torch.float32)
        keep[:] = [ws]
        a.workspace, a.workspace_floats = ptr(ws), n
        return l.wmd_conv_dgrad(C.byref(a), stream)

    choice = (0, 0)
    if tuner.enabled:
        key = "dgrad|%d|%d|%d|%d|%d|%d|%d|%d|%d" % (a.B, a.H, a.W, a.C1, a.up1, a.C2, a.Cout, a.ksize, int(bool(a.dx1)) + 2 * int(bool(a.dx2)))
        if a.ksize == 3 and not a.wp_dgrad_wino:
            key += "|direct"
        choice = tuner.lookup(key)
        if choice is None:
            if torch.cuda.is_current_stream_capturing():
                choice = (0, 0)
            else:
                choice = tuner.tune(key, taps, launch)
    check(launch(*choice), "wmd_conv_dgrad")


def _conv_backward_raw(x1, x2, weight, dz, ksize, pad, up1, has_bias, need_x, need_w, x1_gate=None):
    """dgrad + wgrad through the C ABI for an already-differentiated pre-activation gradient dz."""
    l = _lib.lib()
    s = current_stream()
    B, cout, H, W = dz.shape
    C1 = x1.shape[1]
    C2 = 0 if x2 is None else x2.shape[1]
    dz = _c(dz)
    dx1 = dw = db = None
    if need_x:
        wpd = pack_weights(weight, dgrad=True)
        wpdw = pack_weights_wino(weight, dgrad=True)
        dx1 = torch.empty_like(x1)
        gate_act, gate_slope = (ACT[x1_gate[0]], float(x1_gate[1])) if x1_gate else (0, 0.0)
        a = _lib.ConvDgradArgs(B=B, H=H, W=W, C1=C1, up1=up1, C2=C2, Cout=cout, ksize=ksize, pad_mode=PAD[pad],
                               dz=ptr(dz), wp_dgrad=ptr(wpd), dx1=ptr(dx1), dx2=None, workspace=None, workspace_floats=0,
                               tune_cfg=0, tune_ksplit=0, wp_dgrad_wino=ptr(wpdw),
                               x1_fwd=ptr(x1) if gate_act else None, x1_act=gate_act, x1_slope=gate_slope)
        _dgrad_launch(a, dz.device, 9 if ksize == 3 else 1)
    if need_w:
        dw = torch.empty_like(weight)
        db = torch.empty(cout, device=dz.device, dtype=torch.float32) if has_bias else None
        a = _lib.ConvWgradArgs(B=B, H=H, W=W, C1=C1, up1=up1, C2=C2, Cout=cout, ksize=ksize, pad_mode=PAD[pad],
                               x1=ptr(x1), x2=ptr(x2), dz=ptr(dz), dw=ptr(dw), dbias=ptr(db), workspace=None,
                               workspace_floats=0, tune_cfg=0, tune_nsplit=0)
        _wgrad_launch(a, dz.device)
    return dx1, dw, db


def head3x3(xp, weight_p, bias_p, xn=None, weight_n=None, bias_n=None, pad="reflect", mode=0, scale=1.0, x_gate=None):
    """x_gate: backward-only hint as conv2d_fused's x1_gate (xp / xn are outputs of that activation: their gradients come
    back multiplied by its derivative)."""
    _require_gpu(xp, weight_p, bias_p, xn, weight_n, bias_n)
    return _HeadFn.apply(xp, weight_p, bias_p, xn, weight_n, bias_n, pad, mode, scale, x_gate)


class _StackedHeadsFn(torch.autograd.Function):
    """All wavelet heads of one decoder level in training mode (depth_decoder.py:104-136): 2 (or, with the LL head of the
    coarsest level, 3) chains  Conv1x1 -> LeakyReLU(0.1) -> Conv3x3(refl) -> sigmoid  of the SAME input, combined as
    yh = s (sigma+ - sigma-) [, yl = s_ll sigma_ll].

    Forward: ONE stacked 1x1 GEMM (x read once) + the 3x3 head kernel on channel slices of its output.
    Backward: the heads are handled as one convolution pair with stacked / block-diagonal weights, so every stage is a
    single launch over both (three) heads: the 3x3 weight gradient ([n_out, Ct, 3, 3]: the off-diagonal blocks are
    computed and dropped -- they ride in MFMA rows that would be padding anyway), the 3x3 data gradient (returned already
    multiplied by LeakyReLU'(mid)), the 1x1 weight gradient and the 1x1 data gradient (multiplied by x_gate'(x) when x is
    itself an activation output of the caller).  No per-head dgrad / wgrad / act_bwd launches, no autograd accumulation of
    per-head dx contributions.

    inputs: x, x_gate, scale_hf, scale_ll, then (w1, b1, w3, b3) of the + head, the - head and optionally the LL head."""

    @staticmethod
    def forward(ctx, x, x_gate, scale_hf, scale_ll, *params):
        l = _lib.lib()
        x = _c(x)
        heads = [params[i:i + 4] for i in range(0, len(params), 4)]       # [+, -, (LL)]
        has_ll = len(heads) == 3
        order = ([2] if has_ll else []) + [0, 1]                            # stacked channel order: [LL, +, -]
        w1s, b1s = [heads[k][0] for k in order], [heads[k][1] for k in order]
        mid = conv1x1_stacked_nograd(x, w1s, b1s, act="leaky", slope=0.1)
        offs, o = {}, 0
        for k in order:
            offs[k] = o
            o += heads[k][0].shape[0]
        B, Ct, H, W = mid.shape
        plane = H * W
        base = mid.data_ptr()

        def run(mode, scale, kp, kn=None):
            w3p, b3p = heads[kp][2], heads[kp][3]
            cout = w3p.shape[0]
            y = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
            sp = torch.empty_like(y)
            sn = torch.empty_like(y) if kn is not None else None
            a = _lib.HeadArgs(B=B, H=H, W=W, C=w3p.shape[1], Cout=cout, pad_mode=PAD["reflect"], mode=mode, scale=float(scale),
                              xp=base + 4 * offs[kp] * plane, wgt_p=ptr(_c(w3p.detach())), bias_p=ptr(b3p),
                              xn=None if kn is None else base + 4 * offs[kn] * plane,
                              wgt_n=None if kn is None else ptr(_c(heads[kn][2].detach())),
                              bias_n=None if kn is None else ptr(heads[kn][3]),
                              y=ptr(y), sig_p=ptr(sp), sig_n=ptr(sn), xp_bstride=Ct * plane, xn_bstride=Ct * plane)
            _head_launch(l, a, x.device)
            return y, sp, sn

        yh, sp, sn = run(2, scale_hf, 0, 1)
        yl, sl = None, None
        if has_ll:
            yl, sl, _ = run(1, scale_ll, 2)
        ctx.save_for_backward(x, mid, sp, sn, sl, *params)
        ctx.meta = (x_gate, float(scale_hf), float(scale_ll), has_ll, order, offs)
        ctx.mark_non_differentiable(mid)
        return (yh, yl if has_ll else yh.new_empty(0), mid)

    @staticmethod
    def backward(ctx, d_yh, d_yl, _d_mid):
        x, mid, sp, sn, sl = ctx.saved_tensors[:5]
        dx, grads = _stacked_heads_backward(x, mid, sp, sn, sl, ctx.saved_tensors[5:], ctx.meta, d_yh, d_yl, ctx.needs_input_grad[0])
        return (dx, None, None, None) + tuple(grads)


def _stacked_heads_backward(x, mid, sp, sn, sl, params, meta, d_yh, d_yl, want_dx):
    """Backward of a level's stacked heads from the saved 1x1 outputs `mid` and sigmoid outputs sp / sn / sl (see
    _StackedHeadsFn): -> (dx, [dw1, db1, dw3, db3 of the + head, the - head(, the LL head)])."""
    if True:
        l = _lib.lib()
        x_gate, s_hf, s_ll, has_ll, order, offs = meta
        heads = [params[i:i + 4] for i in range(0, len(params), 4)]
        B, Ct, H, W = mid.shape
        C_in = x.shape[1]
        dev = x.device
        # pre-sigmoid gradients of every head, stacked in the channel order of `mid`: [LL (1)], + (3), - (3)
        parts = []
        if has_ll:
            parts.append(torch.zeros_like(sl) if d_yl is None or d_yl.numel() == 0 else _c(d_yl) * sl * (1.0 - sl) * s_ll)
        d_yh = torch.zeros_like(sp) if d_yh is None else _c(d_yh)      # yh unused by the loss
        parts.append(d_yh * sp * (1.0 - sp) * s_hf)
        parts.append(d_yh * sn * (1.0 - sn) * (-s_hf))
        dy3 = torch.cat(parts, 1)
        n_out = dy3.shape[1]
        rows, r = {}, 0
        for k in order:
            rows[k] = r
            r += heads[k][2].shape[0]
        w1s = torch.cat([heads[k][0].detach() for k in order], 0)
        dzmid = torch.empty_like(mid)
        a3_pending = None
        nsl = sum((heads[k][2].shape[1] + 63) // 64 for k in order)
        # (worth it where a level has pixels to spread: >= 256 wave tiles; the coarsest level stays on the generic kernels, which
        # split its few pixels over channels instead -- tools/head_bwd_microbench.py)
        if _HEAD_BWD and nsl <= 24 and B * H * W >= _HEAD_BWD_MIN_PIXELS and all(heads[k][2].shape[0] in (1, 3) for k in order):
            # the 3x3 stage on its own kernels (wmd_head_bwd.hip): tap-partial rows gathered from dy3, one pass over mid each for
            # the data gradient (returned gated by LeakyReLU'(mid)) and for the weight + bias gradients of every head
            dw3s = {k: torch.empty_like(heads[k][2]) for k in order}
            db3s = {k: torch.empty(heads[k][2].shape[0], device=dev, dtype=torch.float32) for k in order}
            a = _lib.Head3x3BwdArgs(B=B, H=H, W=W, Ct=Ct, n_out=n_out, pad_mode=PAD["reflect"], act=ACT["leaky"], slope=0.1,
                                    dy3=ptr(dy3), mid=ptr(mid), dzmid=ptr(dzmid), n_heads=len(order), workspace=None,
                                    workspace_floats=0)
            keep = []
            for i, k in enumerate(order):
                w3c = _c(heads[k][2].detach())
                keep.append(w3c)
                a.head[i] = _lib.HeadBwdHead(row0=rows[k], nrows=w3c.shape[0], ch0=offs[k], nch=w3c.shape[1], w3=ptr(w3c),
                                             dw3=ptr(dw3s[k]), db3=ptr(db3s[k]))
            n = l.wmd_head3x3_bwd_workspace_floats(C.byref(a))
            ws = torch.empty(max(n, 1), device=dev, dtype=torch.float32)
            a.workspace, a.workspace_floats = ptr(ws), n
            a3_pending = a        # launched below: alone, or merged with the 1x1 stage (wmd_head_bwd)
            wpd1 = None        # packed by the generic 1x1 path on demand
            head_grads3 = lambda k: (dw3s[k], db3s[k])
        else:
            # block-diagonal 3x3 filter [n_out, Ct, 3, 3]
            w3bd = torch.zeros((n_out, Ct, 3, 3), device=dev, dtype=torch.float32)
            for k in order:
                w3 = heads[k][2].detach()
                w3bd[rows[k]:rows[k] + w3.shape[0], offs[k]:offs[k] + w3.shape[1]] = w3
            # 3x3: weight gradient of the stacked filter (diagonal blocks are the heads' gradients) ...
            dw3f = torch.empty_like(w3bd)
            db3f = torch.empty(n_out, device=dev, dtype=torch.float32)
            a = _lib.ConvWgradArgs(B=B, H=H, W=W, C1=Ct, up1=1, C2=0, Cout=n_out, ksize=3, pad_mode=PAD["reflect"], x1=ptr(mid), x2=None,
                                   dz=ptr(dy3), dw=ptr(dw3f), dbias=ptr(db3f), workspace=None, workspace_floats=0, tune_cfg=0,
                                   tune_nsplit=0)
            _wgrad_launch(a, dev)
            # ... and data gradient, gated by LeakyReLU'(mid): dzmid
            # the three weight images of this backward (3x3 data gradient direct + Winograd, 1x1 data gradient) in one launch
            imgs = pack_many([(w3bd, ("dgrad", "wino_dgrad") if _WINOGRAD else ("dgrad",))] + ([(w1s, ("dgrad",))] if want_dx else []))
            wpd, wpdw = imgs[0]["dgrad"], imgs[0].get("wino_dgrad")
            a = _lib.ConvDgradArgs(B=B, H=H, W=W, C1=Ct, up1=1, C2=0, Cout=n_out, ksize=3, pad_mode=PAD["reflect"], dz=ptr(dy3),
                                   wp_dgrad=ptr(wpd), dx1=ptr(dzmid), dx2=None, workspace=None, workspace_floats=0, tune_cfg=0,
                                   tune_ksplit=0, wp_dgrad_wino=ptr(wpdw), x1_fwd=ptr(mid), x1_act=ACT["leaky"], x1_slope=0.1)
            _dgrad_launch(a, dev, 9)
            wpd1 = imgs[1]["dgrad"] if want_dx else None
            head_grads3 = lambda k: (dw3f[rows[k]:rows[k] + heads[k][2].shape[0], offs[k]:offs[k] + heads[k][2].shape[1]].contiguous(),
                                     db3f[rows[k]:rows[k] + heads[k][2].shape[0]])
        dw1f = torch.empty_like(w1s)
        db1f = torch.empty(Ct, device=dev, dtype=torch.float32)
        dx = torch.empty_like(x) if want_dx else None
        gate_act, gate_slope = (ACT[x_gate[0]], float(x_gate[1])) if x_gate else (0, 0.0)
        # (alone they only win at the finest level; as components of the merged second-stage launch -- a level without the
        # low-pass head whose 3x3 stage runs on its own kernels too -- wherever those do)
        merged = a3_pending is not None and not has_ll and _HEAD_BWD_MERGED
        if _HEAD_BWD and B * H * W >= (_HEAD_BWD_MIN_PIXELS if merged else _HEAD_BWD1_MIN_PIXELS) and Ct % 8 == 0 and \
                gate_act in (ACT["none"], ACT["leaky"], ACT["elu"]):
            # 1x1 stage on its own kernels (wmd_head_bwd1.hip): one pass over dz and x each for dx (gated by the caller's
            # activation if x has one) and for the stacked weight + bias gradient
            w1c = _c(w1s.reshape(Ct, C_in))
            a = _lib.Head1x1BwdArgs(B=B, H=H, W=W, C=C_in, Ct=Ct, x_act=gate_act, x_slope=gate_slope, dz=ptr(dzmid), x=ptr(x),
                                    w1=ptr(w1c), dx=ptr(dx), dw1=ptr(dw1f), db1=ptr(db1f), workspace=None, workspace_floats=0)
            n = l.wmd_head1x1_bwd_workspace_floats(C.byref(a))
            ws1 = torch.empty(max(n, 1), device=dev, dtype=torch.float32)
            a.workspace, a.workspace_floats = ptr(ws1), n
            if merged:
                # three launches for both stages: 3x3 data gradient, then [3x3 weights | 1x1 data | 1x1 weights] as one, then
                # both reduces as one
                check(l.wmd_head_bwd(C.byref(a3_pending), C.byref(a), current_stream()), "wmd_head_bwd")
                a3_pending = None
            else:
                if a3_pending is not None:
                    check(l.wmd_head3x3_bwd(C.byref(a3_pending), current_stream()), "wmd_head3x3_bwd")
                    a3_pending = None
                check(l.wmd_head1x1_bwd(C.byref(a), current_stream()), "wmd_head1x1_bwd")
        else:
            if a3_pending is not None:
                check(l.wmd_head3x3_bwd(C.byref(a3_pending), current_stream()), "wmd_head3x3_bwd")
                a3_pending = None
            # 1x1: weight gradient of the stacked filter ...
            a = _lib.ConvWgradArgs(B=B, H=H, W=W, C1=C_in, up1=1, C2=0, Cout=Ct, ksize=1, pad_mode=PAD["zero"], x1=ptr(x), x2=None,
                                   dz=ptr(dzmid), dw=ptr(dw1f), dbias=ptr(db1f), workspace=None, workspace_floats=0, tune_cfg=0,
                                   tune_nsplit=0)
            _wgrad_launch(a, dev)
            # ... and data gradient (one GEMM over all heads' mid channels), gated by the caller's activation if x has one
            if want_dx:
                if wpd1 is None:
                    wpd1 = pack_many([(w1s, ("dgrad",))])[0]["dgrad"]
                a = _lib.ConvDgradArgs(B=B, H=H, W=W, C1=C_in, up1=1, C2=0, Cout=Ct, ksize=1, pad_mode=PAD["zero"], dz=ptr(dzmid),
                                       wp_dgrad=ptr(wpd1), dx1=ptr(dx), dx2=None, workspace=None, workspace_floats=0, tune_cfg=0,
                                       tune_ksplit=0, wp_dgrad_wino=None, x1_fwd=ptr(x) if gate_act else None, x1_act=gate_act,
                                       x1_slope=gate_slope)
                _dgrad_launch(a, dev, 1)
        grads = []
        for k in range(len(heads)):
            w1, _b1, w3, _b3 = heads[k]
            c1, c3 = w1.shape[0], w3.shape[0]
            dw3k, db3k = head_grads3(k)
            grads += [dw1f[offs[k]:offs[k] + c1].reshape(w1.shape), db1f[offs[k]:offs[k] + c1], dw3k, db3k]
        return dx, grads


def stacked_heads(x, head_p, head_n, scale_hf, head_ll=None, scale_ll=1.0, x_gate=None, return_mid=False):
    """Training-mode wavelet heads of one level (see _StackedHeadsFn).  head_* = (w1, b1, w3, b3).
    Returns (yh [B,3,H,W], yl [B,1,H,W] or None)."""
    params = tuple(head_p) + tuple(head_n) + (tuple(head_ll) if head_ll is not None else ())
    _require_gpu(x, *params)
    yh, yl, mid = _StackedHeadsFn.apply(x, x_gate, scale_hf, scale_ll, *params)
    if return_mid:   # [B, (C/4 +) 2C, H, W]: LeakyReLU outputs in the order [LL], +, - (diagnostics; carries no gradient)
        return yh, (yl if head_ll is not None else None), mid
    return yh, (yl if head_ll is not None else None)


# ---------------------------------------------------------------------------------------------
# inference-only fast paths (no autograd): stacked 1x1 heads + head3x3 on channel slices
# ---------------------------------------------------------------------------------------------

def stacked_pack(weights, biases):
    """Packed image + bias of several [Cout_k, Cin, 1, 1] filters stacked along Cout (memoised on weights[0])."""
    couts = [w.shape[0] for w in weights]
    if any(c % 16 for c in couts[:-1]):
        raise _lib.WmdError("stacked heads need out-channel counts that are multiples of 16")
    tag = tuple((w._version, w.data_ptr(), b._version, b.data_ptr()) for w, b in zip(weights, biases)) + (_pack_generation[0],)
    hit = getattr(weights[0], "_wmd_pack_stack", None) if _PACK_CACHE else None
    if hit is not None and hit[0] == tag:
        return hit[1], hit[2]
    _imgs, wp = pack_many([(w, ("fwd",)) for w in weights], with_buffer=True)   # consecutive slices = the stacked image
    bias = torch.cat([b.detach() for b in biases])
    if _PACK_CACHE and not torch.cuda.is_current_stream_capturing():
        try:
            weights[0]._wmd_pack_stack = (tag, wp, bias)
        except AttributeError:
            pass
    return wp, bias


def conv1x1_stacked_nograd(x, weights, biases, act="leaky", slope=0.1):
    """Several 1x1 convolutions of the SAME input as one launch: their packed weight images are simply
    concatenated along the out-channel-tile axis (every Cout is a multiple of 16), so x is read once.
    Returns [B, sum(Cout_k), H, W]."""
    x = _c(x)
    couts = [w.shape[0] for w in weights]
    wp, bias = stacked_pack(weights, biases)
    return _conv_fwd_raw(x, None, wp, bias, sum(couts), 1, "zero", act, slope, 1)


def head3x3_nograd(x_full, cin, off_p, weight_p, bias_p, off_n=None, weight_n=None, bias_n=None, pad="reflect", mode=0,
                   scale=1.0):
    """head3x3 reading its cin-channel input(s) as channel slices [off, off+cin) of one [B,Ctot,H,W] tensor."""
    l = _lib.lib()
    B, Ctot, H, W = x_full.shape
    cout = weight_p.shape[0]
    y = torch.empty((B, cout, H, W), device=x_full.device, dtype=torch.float32)
    plane = H * W
    base = x_full.data_ptr()
    a = _lib.HeadArgs(B=B, H=H, W=W, C=cin, Cout=cout, pad_mode=PAD[pad], mode=mode, scale=float(scale),
                      xp=base + 4 * off_p * plane, wgt_p=ptr(_c(weight_p.detach())), bias_p=ptr(bias_p),
                      xn=None if off_n is None else base + 4 * off_n * plane,
                      wgt_n=None if weight_n is None else ptr(_c(weight_n.detach())), bias_n=ptr(bias_n),
                      y=ptr(y), sig_p=None, sig_n=None, xp_bstride=Ctot * plane, xn_bstride=Ctot * plane)
    _head_launch(l, a, x_full.device)
    return y


def _tap_partial_pack(w3p, w3n):
    """[3,C,3,3] x 2 -> two packed [27,C,1,1] images (row co*9+tap), memoised on w3p."""
    tag = (w3p._version, w3p.data_ptr(), w3n._version, w3n.data_ptr(), _pack_generation[0])
    hit = getattr(w3p, "_wmd_pack_t27", None) if _PACK_CACHE else None
    if hit is not None and hit[0] == tag:
        return hit[1]
    cin = w3p.shape[1]
    w27s = [w.detach().permute(0, 2, 3, 1).reshape(27, cin, 1, 1).contiguous() for w in (w3p, w3n)]
    _imgs, wp = pack_many([(w, ("fwd",)) for w in w27s], with_buffer=True)
    if _PACK_CACHE and not torch.cuda.is_current_stream_capturing():
        try:
            w3p._wmd_pack_t27 = (tag, wp)
        except AttributeError:
            pass
    return wp


def _ll_chain_pack(w1l, w3l):
    """Weight images of the low-pass chain for wmd_head_fused_fwd(chain = 1): the [C/4,C,1,1] filter as it is, and the
    [1,C/4,3,3] filter as a [27,C/4,1,1] image whose rows 0..8 are its nine taps (rows 9..26 zero).  Memoised on w1l."""
    tag = (w1l._version, w1l.data_ptr(), w3l._version, w3l.data_ptr(), _pack_generation[0])
    hit = getattr(w1l, "_wmd_pack_ll", None) if _PACK_CACHE else None
    if hit is not None and hit[0] == tag:
        return hit[1], hit[2]
    cm = w3l.shape[1]
    w27 = torch.zeros((27, cm, 1, 1), device=w3l.device, dtype=torch.float32)
    w27[:9, :, 0, 0] = w3l.detach()[0].permute(1, 2, 0).reshape(9, cm)
    imgs = pack_many([(w1l, ("fwd",)), (w27, ("fwd",))])
    wp1, wp2 = imgs[0]["fwd"], imgs[1]["fwd"]
    if _PACK_CACHE and not torch.cuda.is_current_stream_capturing():
        try:
            w1l._wmd_pack_ll = (tag, wp1, wp2)
        except AttributeError:
            pass
    return wp1, wp2


FUSED_HEAD_WIDTHS = (32, 64, 128, 256)
_HEAD_BWD = os.environ.get("WMD_HEAD_BWD", "1") != "0"                 # 0: 3x3 head backward on the generic dgrad / wgrad kernels
_HEAD_BWD_MERGED = os.environ.get("WMD_HEAD_BWD_MERGED", "1") != "0"   # 0: wmd_head3x3_bwd + wmd_head1x1_bwd as separate launch sets
_HEAD_BWD_MIN_PIXELS = int(os.environ.get("WMD_HEAD_BWD_MIN_PIXELS", "16384"))
_HEAD_BWD1_MIN_PIXELS = int(os.environ.get("WMD_HEAD_BWD1_MIN_PIXELS", "196608"))     # same for the 1x1 stage (wmd_head1x1_bwd):
# its kernels walk the whole channel sum per 64-pixel wave tile and only win where a level has thousands of tiles (the finest one)
_TWO_LAUNCH_HEAD = os.environ.get("WMD_TWO_LAUNCH_HEAD", "0") == "1"   # development switch: A/B the two forms
_LL_FOLD = os.environ.get("WMD_LL_FOLD", "1") != "0"                   # 0: the low-pass head on its own three launches
_LL_MERGE = os.environ.get("WMD_LL_MERGE", "1") != "0"                 # 0: the low-pass chain as a launch of its own (round 3)


def head_level_folds_range_keys(C_):
    """True when head_fused_level_nograd(range_keys=...) at this width runs the two-launch form, whose second launch maintains
    the keys (the one-launch kernel of the finest level has no level after it that could want them)."""
    return not (bool(_lib.lib().wmd_head_level_supported(int(C_))) and not _TWO_LAUNCH_HEAD)


def head_fused_level_nograd(x, head_p, head_n, scale, yl=None, disp_scale=None, clamp01=False, head_ll=None, scale_ll=1.0,
                            yh_mask=None, run_mask=None, range_keys=None, train=None):
    """Inference form of one level's high-frequency heads + (optionally) the Haar IDWT in one launch (C = 32:
    wmd_head_level_fwd, every intermediate in LDS) or two: wmd_head_fused_fwd (1x1 -> LeakyReLU -> 27 tap-partials per
    side, intermediate stays on chip) and wmd_head_shiftsum_fwd (9-tap gather, bias, sigmoid, combine, IDWT).
    head_* = (w1, b1, w3, b3).  head_ll: the coarsest level's low-pass head (C -> C/4 -> 1, sigmoid * scale_ll); at C = 256
    it is a small third launch of the same fused kernel (tap-partials into planes 54..62 of the shared buffer) that the
    shift-sum completes and feeds to the synthesis as its low-pass input (yl must be None); other widths: own operators.
    yh_mask (uint8 [B,H,W]): yh is zeroed outside it before the store and the synthesis (depth_decoder.py:272).
    run_mask (uint8 [B,H,W], two-launch form): pixel runs without a set byte skip the GEMMs (wmd_head_fused_args.run_mask);
    range_keys (int32 [B,2], two-launch form): the (min, max) of the new low-pass plane are folded into it by the second
    launch's epilogue (wmd_head_shiftsum_args.range_keys) -- returns None for them when the one-launch kernel ran.
    train (round 5: the training forward on these kernels, _FusedLevelFn): dict(mid=[B,Ct,H,W], off_p=, off_n=, off_ll=,
    sig_p=, sig_n= [B,3,H,W], sig_ll= [B,1,H,W] or None) -- the LeakyReLU outputs of the 1x1 stage and the sigmoid outputs are
    written there as well (what autograd would have kept for the backward).
    Returns (yh [B,1,3,H,W], out or None, disp or None[, yl_ll [B,1,H,W] when head_ll is given])."""
    l = _lib.lib()
    x = _c(x)
    B, Cc, H, W = x.shape
    tr = train or {}
    if train is not None and not fused_train_supported(Cc, H, W, head_ll is not None):
        raise _lib.WmdError("head_fused_level_nograd: no fused training forward for C=%d, %dx%d%s" % (Cc, H, W, " + LL" if head_ll is not None else ""))
    if yh_mask is not None and (yh_mask.dtype != torch.uint8 or yh_mask.numel() != B * H * W or not yh_mask.is_contiguous()):
        raise _lib.WmdError("head_fused_level_nograd: yh_mask must be a contiguous uint8 [B,H,W] tensor")
    (w1p, b1p, w3p, b3p), (w1n, b1n, w3n, b3n) = head_p, head_n
    one_launch = bool(l.wmd_head_level_supported(Cc)) and not _TWO_LAUNCH_HEAD
    yl_ll = None
    if head_ll is not None:
        if yl is not None:
            raise _lib.WmdError("head_fused_level_nograd: head_ll supplies the low-pass input; yl must be None")
        w1l, b1l, w3l, b3l = head_ll
        if one_launch or Cc != 256 or not _LL_FOLD:     # the low-pass head on its own operators
            mid0 = conv2d_fused(x, w1l, b1l, pad="zero", act="leaky", slope=0.1)
            yl = yl_ll = head3x3(mid0, w3l, b3l, pad="reflect", mode=1, scale=scale_ll)
            head_ll = None
    wp1, bias1 = stacked_pack([w1p, w1n], [b1p, b1n])
    wp2 = _tap_partial_pack(w3p, w3n)
    s = current_stream()
    yh = torch.empty((B, 3, H, W), device=x.device, dtype=torch.float32)
    out = disp = None
    if yl is not None or head_ll is not None:
        yl = _c(yl) if yl is not None else None
        out = torch.empty((B, 1, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        disp = torch.empty_like(out) if disp_scale is not None else None
    if one_launch:
        # one launch: every intermediate of the level stays in LDS
        a = _lib.HeadLevelArgs(B=B, H=H, W=W, C=Cc, pad_mode=PAD["reflect"], slope=0.1, scale=float(scale), x=ptr(x),
                               wp1=ptr(wp1), bias1=ptr(bias1), wp2=ptr(wp2), bias_p=ptr(b3p), bias_n=ptr(b3n), yh=ptr(yh),
                               yl=ptr(yl), out=ptr(out), disp=ptr(disp), disp_scale=float(disp_scale or 1.0),
                               clamp01=int(clamp01), yh_mask=ptr(yh_mask))
        if train is not None:
            a.mid_out, a.mid_ct, a.mid_off_p, a.mid_off_n = ptr(tr["mid"]), tr["mid"].shape[1], tr["off_p"], tr["off_n"]
            a.sig_p, a.sig_n = ptr(tr["sig_p"]), ptr(tr["sig_n"])
        check(l.wmd_head_level_fwd(C.byref(a), s), "wmd_head_level_fwd")
    else:
        planes = 81 if head_ll is not None else 54
        t = torch.empty((B, planes, H, W), device=x.device, dtype=torch.float32)
        a = _lib.HeadFusedArgs(B=B, H=H, W=W, C=Cc, slope=0.1, x=ptr(x), wp1=ptr(wp1), bias1=ptr(bias1), wp2=ptr(wp2), t=ptr(t),
                               chain=0, t_planes=planes, run_mask=ptr(run_mask))
        if head_ll is not None:     # the low-pass chain rides in the same call (planes 54..62): a third group of workgroups of
            wpl1, wpl2 = _ll_chain_pack(w1l, w3l)       # the chained kernel's launch, or a second launch issued by the library
            yl_ll = torch.empty((B, 1, H, W), device=x.device, dtype=torch.float32)
            b1l_c = _c(b1l.detach())
            if _LL_MERGE:
                a.ll_wp1, a.ll_bias1, a.ll_wp2 = ptr(wpl1), ptr(b1l_c), ptr(wpl2)
        if train is not None:
            a.mid_out, a.mid_ct, a.mid_off_p, a.mid_off_n = ptr(tr["mid"]), tr["mid"].shape[1], tr["off_p"], tr["off_n"]
            a.mid_off_ll = tr.get("off_ll", 0)
        check(l.wmd_head_fused_fwd(C.byref(a), s), "wmd_head_fused_fwd")
        if head_ll is not None and not _LL_MERGE:
            a = _lib.HeadFusedArgs(B=B, H=H, W=W, C=Cc, slope=0.1, x=ptr(x), wp1=ptr(wpl1), bias1=ptr(b1l_c), wp2=ptr(wpl2),
                                   t=ptr(t), chain=1, t_planes=planes)
            check(l.wmd_head_fused_fwd(C.byref(a), s), "wmd_head_fused_fwd")
        g = _lib.HeadShiftsumArgs(B=B, H=H, W=W, pad_mode=PAD["reflect"], scale=float(scale), t=ptr(t), bias_p=ptr(b3p),
                                  bias_n=ptr(b3n), yh=ptr(yh), yl=ptr(yl), out=ptr(out), disp=ptr(disp),
                                  disp_scale=float(disp_scale or 1.0), clamp01=int(clamp01),
                                  bias_ll=ptr(b3l) if head_ll is not None else None, scale_ll=float(scale_ll),
                                  yl_out=ptr(yl_ll) if head_ll is not None else None, yh_mask=ptr(yh_mask),
                                  range_keys=ptr(range_keys) if out is not None else None)
        if train is not None:
            g.sig_p, g.sig_n = ptr(tr["sig_p"]), ptr(tr["sig_n"])
            g.sig_ll = ptr(tr.get("sig_ll")) if head_ll is not None else None
        check(l.wmd_head_shiftsum_fwd(C.byref(g), s), "wmd_head_shiftsum_fwd")
    if yl_ll is not None:
        return yh.unsqueeze(1), out, disp, yl_ll
    return yh.unsqueeze(1), out, disp


_SHIFTSUM_CHAIN = os.environ.get("WMD_SHIFTSUM_CHAIN", "1") != "0"   # 0: one wmd_head_shiftsum_fwd launch per level
# ... up to this many pixels at the finest chained level: the single launch removes two graph nodes from a latency-bound forward (one
# frame 640x192: 0.221 -> 0.211 ms) but completes levels 4 and 3 from planes that have left the caches by then -- at batch 12
# (92 160 pixels) the step measured 0.5 - 1 % slower with it (0.588 / 0.592 vs 0.585 / 0.584 ms medians, same box)
_SHIFTSUM_CHAIN_MAX_PIXELS = int(os.environ.get("WMD_SHIFTSUM_CHAIN_MAX_PIXELS", "32768"))


def shiftsum_chain_supported(widths, finest_pixels=0):
    """Dense inference: can the levels of these head widths (coarse to fine) run as chained GEMM launches + ONE completion launch
    (head_fused_gemm_nograd + head_shiftsum_chain_nograd)?  Every one must be a two-launch width (the one-launch kernel of C = 32
    completes itself) and the coarsest low-pass head must ride in the C = 256 launch; finest_pixels = B*H*W of the last level
    (the launch pays for latency-bound forwards only: _SHIFTSUM_CHAIN_MAX_PIXELS)."""
    l = _lib.lib()
    return (_SHIFTSUM_CHAIN and finest_pixels <= _SHIFTSUM_CHAIN_MAX_PIXELS and not _TWO_LAUNCH_HEAD and 1 <= len(widths) <= 3 and _LL_FOLD and _LL_MERGE and _HEAD_CHAIN_ON and
            all(int(c) in (64, 128, 256) and not l.wmd_head_level_supported(int(c)) for c in widths) and int(widths[0]) == 256)


def head_fused_gemm_multi_nograd(levels):
    """The first stage of up to three levels in ONE launch (round 6, wmd_head_fused_multi_fwd): levels = [(x, head_p, head_n,
    head_ll or None), ...] coarse to fine -> the same items head_fused_gemm_nograd returns, one per level (bit-identical planes).
    The coarse levels cannot balance 256 CUs alone; together they do (profiles/r06_notes.md section 6)."""
    l = _lib.lib()
    n = len(levels)
    arr = (_lib.HeadFusedArgs * n)()
    items = []
    for k, (x, head_p, head_n, head_ll) in enumerate(levels):
        x = _c(x)
        B, Cc, H, W = x.shape
        (w1p, b1p, w3p, b3p), (w1n, b1n, w3n, b3n) = head_p, head_n
        wp1, bias1 = stacked_pack([w1p, w1n], [b1p, b1n])
        wp2 = _tap_partial_pack(w3p, w3n)
        planes = 81 if head_ll is not None else 54
        t = torch.empty((B, planes, H, W), device=x.device, dtype=torch.float32)
        arr[k] = _lib.HeadFusedArgs(B=B, H=H, W=W, C=Cc, slope=0.1, x=ptr(x), wp1=ptr(wp1), bias1=ptr(bias1), wp2=ptr(wp2), t=ptr(t),
                                    chain=0, t_planes=planes)
        keep = [x, wp1, bias1, wp2]
        b3l = None
        if head_ll is not None:
            w1l, b1l, w3l, b3l = head_ll
            wpl1, wpl2 = _ll_chain_pack(w1l, w3l)
            b1l_c = _c(b1l.detach())
            arr[k].ll_wp1, arr[k].ll_bias1, arr[k].ll_wp2 = ptr(wpl1), ptr(b1l_c), ptr(wpl2)
            keep += [wpl1, wpl2, b1l_c]
        items.append(dict(t=t, B=B, H=H, W=W, b3p=b3p, b3n=b3n, b3l=b3l, has_ll=head_ll is not None, keep=keep))
    check(l.wmd_head_fused_multi_fwd(arr, n, current_stream()), "wmd_head_fused_multi_fwd")
    return items


def head_shiftsum_item_nograd(item, scale, disp_scale, yl=None, scale_ll=1.0, clamp01=True):
    """Completes ONE level from an item of head_fused_gemm[_multi]_nograd (wmd_head_shiftsum_fwd): -> (yh [B,1,3,H,W], out, disp,
    yl_ll or None).  yl: the level's low-pass input (None for the level whose item carries the low-pass head)."""
    l = _lib.lib()
    B, H, W = item["B"], item["H"], item["W"]
    dev = item["t"].device
    yh = torch.empty((B, 3, H, W), device=dev, dtype=torch.float32)
    out = torch.empty((B, 1, 2 * H, 2 * W), device=dev, dtype=torch.float32)
    disp = torch.empty_like(out)
    yl_ll = torch.empty((B, 1, H, W), device=dev, dtype=torch.float32) if item["has_ll"] else None
    if yl_ll is None and yl is None:
        raise _lib.WmdError("head_shiftsum_item_nograd: the level needs its low-pass input (yl) or the low-pass head")
    yl_c = _c(yl) if (yl is not None and yl_ll is None) else None
    g = _lib.HeadShiftsumArgs(B=B, H=H, W=W, pad_mode=PAD["reflect"], scale=float(scale), t=ptr(item["t"]), bias_p=ptr(item["b3p"]),
                              bias_n=ptr(item["b3n"]), yh=ptr(yh), yl=ptr(yl_c), out=ptr(out), disp=ptr(disp), disp_scale=float(disp_scale),
                              clamp01=int(clamp01), bias_ll=ptr(item["b3l"]) if yl_ll is not None else None, scale_ll=float(scale_ll),
                              yl_out=ptr(yl_ll))
    check(l.wmd_head_shiftsum_fwd(C.byref(g), current_stream()), "wmd_head_shiftsum_fwd")
    return yh.unsqueeze(1), out, disp, yl_ll


_HEAD_CHAIN_MULTI = os.environ.get("WMD_HEAD_CHAIN_MULTI", "1") != "0"   # 0: every level's first stage as a launch of its own


def head_chain_multi_supported(widths):
    """Dense inference: may the first stages of these levels (coarse to fine) be postponed and run as ONE launch?"""
    l = _lib.lib()
    return (_HEAD_CHAIN_MULTI and not _TWO_LAUNCH_HEAD and 2 <= len(widths) <= 3 and _LL_FOLD and _LL_MERGE and _HEAD_CHAIN_ON and
            all(int(c) in (64, 128, 256) and not l.wmd_head_level_supported(int(c)) for c in widths))


def head_fused_gemm_nograd(x, head_p, head_n, head_ll=None):
    """First launch of the two-launch head form alone (wmd_head_fused_fwd: 1x1 -> LeakyReLU -> tap-partial planes, the coarsest
    level's low-pass chain riding along): -> what head_shiftsum_chain_nograd needs to complete the level later."""
    l = _lib.lib()
    x = _c(x)
    B, Cc, H, W = x.shape
    (w1p, b1p, w3p, b3p), (w1n, b1n, w3n, b3n) = head_p, head_n
    wp1, bias1 = stacked_pack([w1p, w1n], [b1p, b1n])
    wp2 = _tap_partial_pack(w3p, w3n)
    planes = 81 if head_ll is not None else 54
    t = torch.empty((B, planes, H, W), device=x.device, dtype=torch.float32)
    a = _lib.HeadFusedArgs(B=B, H=H, W=W, C=Cc, slope=0.1, x=ptr(x), wp1=ptr(wp1), bias1=ptr(bias1), wp2=ptr(wp2), t=ptr(t),
                           chain=0, t_planes=planes)
    keep = [x, wp1, bias1, wp2]
    b3l = None
    if head_ll is not None:
        w1l, b1l, w3l, b3l = head_ll
        wpl1, wpl2 = _ll_chain_pack(w1l, w3l)
        b1l_c = _c(b1l.detach())
        a.ll_wp1, a.ll_bias1, a.ll_wp2 = ptr(wpl1), ptr(b1l_c), ptr(wpl2)
        keep += [wpl1, wpl2, b1l_c]
    check(l.wmd_head_fused_fwd(C.byref(a), current_stream()), "wmd_head_fused_fwd")
    return dict(t=t, B=B, H=H, W=W, b3p=b3p, b3n=b3n, b3l=b3l, has_ll=head_ll is not None, keep=keep)


def head_shiftsum_chain_nograd(items, scales, disp_scales, scale_ll=1.0, yl=None, clamp01=True):
    """Completes up to three consecutive levels (coarse to fine; items from head_fused_gemm_nograd) in ONE launch
    (wmd_head_shiftsum_chain_fwd): per level -> (yh [B,1,3,H,W], out [B,1,2H,2W], disp, yl_ll or None)."""
    l = _lib.lib()
    n = len(items)
    arr = (_lib.HeadShiftsumArgs * n)()
    res = []
    yl_c = _c(yl) if yl is not None else None   # held until after the launch: a dropped temporary's block could be handed to yh / out / disp below
    for k, it in enumerate(items):
        B, H, W = it["B"], it["H"], it["W"]
        dev = it["t"].device
        yh = torch.empty((B, 3, H, W), device=dev, dtype=torch.float32)
        out = torch.empty((B, 1, 2 * H, 2 * W), device=dev, dtype=torch.float32)
        disp = torch.empty_like(out)
        yl_ll = torch.empty((B, 1, H, W), device=dev, dtype=torch.float32) if (k == 0 and it["has_ll"]) else None
        if k == 0 and not it["has_ll"] and yl is None:
            raise _lib.WmdError("head_shiftsum_chain_nograd: the first level needs its low-pass input (yl) or the low-pass head")
        arr[k] = _lib.HeadShiftsumArgs(B=B, H=H, W=W, pad_mode=PAD["reflect"], scale=float(scales[k]), t=ptr(it["t"]), bias_p=ptr(it["b3p"]),
                                       bias_n=ptr(it["b3n"]), yh=ptr(yh), yl=ptr(yl_c) if (k == 0 and yl_ll is None) else None, out=ptr(out),
                                       disp=ptr(disp), disp_scale=float(disp_scales[k]), clamp01=int(clamp01),
                                       bias_ll=ptr(it["b3l"]) if yl_ll is not None else None, scale_ll=float(scale_ll),
                                       yl_out=ptr(yl_ll))
        res.append((yh.unsqueeze(1), out, disp, yl_ll))
    check(l.wmd_head_shiftsum_chain_fwd(arr, n, current_stream()), "wmd_head_shiftsum_chain_fwd")
    del yl_c   # (the items' `keep` lists -- contiguous copies the GEMM launches read -- live in `items` until the caller drops them)
    return res


_HEAD_PYRAMID = os.environ.get("WMD_HEAD_PYRAMID", "1") != "0"   # 0: the coarser levels' completions as a launch of their own


def head_level_pyramid_supported(C_, B, H, W):
    """Dense inference: can the C = 32 level run its heads + synthesis AND the coarser levels' completions in one launch?"""
    return _HEAD_PYRAMID and not _TWO_LAUNCH_HEAD and os.environ.get("WMD_HEAD_STREAM", "1") != "0" and \
        _lib.lib().wmd_head_level_pyramid_supported(int(C_), int(B), int(H), int(W)) >= (1 if os.environ.get("WMD_HEAD_PYRAMID") == "2" else 2)


def head_level_pyramid_nograd(x, head_p, head_n, scale, disp_scale, items, scales, disp_scales, scale_ll=1.0, yl=None, clamp01=True):
    """Round 6 (wmd_head_level_pyramid_fwd): the finest level's heads + synthesis AND the completions of the coarser levels (items of
    head_fused_gemm[_multi]_nograd, coarse to fine) in ONE launch -- the streaming kernel's epilogue waves complete the coarser
    levels over each unit's footprint first and hand the low-pass tiles down through LDS.  -> ([(yh, out, disp, yl_ll or None) per
    coarse level], (yh, out, disp) of this level); same bits as head_shiftsum_chain_nograd + head_fused_level_nograd."""
    l = _lib.lib()
    x = _c(x)
    B, Cc, H, W = x.shape
    n = len(items)
    arr = (_lib.HeadShiftsumArgs * n)()
    res = []
    yl_c = _c(yl) if yl is not None else None
    for k, it in enumerate(items):
        Bk, Hk, Wk = it["B"], it["H"], it["W"]
        dev = it["t"].device
        yh = torch.empty((Bk, 3, Hk, Wk), device=dev, dtype=torch.float32)
        out = torch.empty((Bk, 1, 2 * Hk, 2 * Wk), device=dev, dtype=torch.float32)
        disp = torch.empty_like(out)
        yl_ll = torch.empty((Bk, 1, Hk, Wk), device=dev, dtype=torch.float32) if (k == 0 and it["has_ll"]) else None
        if k == 0 and not it["has_ll"] and yl is None:
            raise _lib.WmdError("head_level_pyramid_nograd: the first coarse level needs its low-pass input (yl) or the low-pass head")
        arr[k] = _lib.HeadShiftsumArgs(B=Bk, H=Hk, W=Wk, pad_mode=PAD["reflect"], scale=float(scales[k]), t=ptr(it["t"]), bias_p=ptr(it["b3p"]),
                                       bias_n=ptr(it["b3n"]), yh=ptr(yh), yl=ptr(yl_c) if (k == 0 and yl_ll is None) else None, out=ptr(out),
                                       disp=ptr(disp), disp_scale=float(disp_scales[k]), clamp01=int(clamp01),
                                       bias_ll=ptr(it["b3l"]) if yl_ll is not None else None, scale_ll=float(scale_ll),
                                       yl_out=ptr(yl_ll))
        res.append((yh.unsqueeze(1), out, disp, yl_ll))
    (w1p, b1p, w3p, b3p), (w1n, b1n, w3n, b3n) = head_p, head_n
    wp1, bias1 = stacked_pack([w1p, w1n], [b1p, b1n])
    wp2 = _tap_partial_pack(w3p, w3n)
    yh1 = torch.empty((B, 3, H, W), device=x.device, dtype=torch.float32)
    out1 = torch.empty((B, 1, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    disp1 = torch.empty_like(out1) if disp_scale is not None else None
    a = _lib.HeadLevelArgs(B=B, H=H, W=W, C=Cc, pad_mode=PAD["reflect"], slope=0.1, scale=float(scale), x=ptr(x),
                           wp1=ptr(wp1), bias1=ptr(bias1), wp2=ptr(wp2), bias_p=ptr(b3p), bias_n=ptr(b3n), yh=ptr(yh1),
                           yl=None, out=ptr(out1), disp=ptr(disp1), disp_scale=float(disp_scale or 1.0), clamp01=int(clamp01), yh_mask=None)
    check(l.wmd_head_level_pyramid_fwd(C.byref(a), arr, n, current_stream()), "wmd_head_level_pyramid_fwd")
    del yl_c
    return res, (yh1.unsqueeze(1), out1, disp1)


_TRAIN_FUSED = os.environ.get("WMD_TRAIN_FUSED_HEADS", "1") != "0"   # 0: training forward of the heads on _StackedHeadsFn + idwt_haar
_HEAD_CHAIN_ON = os.environ.get("WMD_HEAD_CHAIN", "1") != "0"


def fused_train_supported(C_, H, W, has_ll):
    """Can the training forward of a level's heads run on the fused inference kernels (_FusedLevelFn)?  They must be able to
    write the 1x1 outputs: the one-launch kernel (C = 32, no low-pass head) or the chained kernel (C = 64 / 128 / 256, planes of
    a multiple of 4 pixels; the low-pass head only folded into the C = 256 launch)."""
    if not _TRAIN_FUSED or _TWO_LAUNCH_HEAD:
        return False
    if bool(_lib.lib().wmd_head_level_supported(int(C_))):
        return not has_ll
    if not _HEAD_CHAIN_ON or int(C_) not in (64, 128, 256) or (H * W) % 4:
        return False
    return (not has_ll) or (int(C_) == 256 and _LL_FOLD and _LL_MERGE)


class _FusedLevelFn(torch.autograd.Function):
    """One decoder level's wavelet heads AND its Haar synthesis in training mode on the inference kernels (round 5; VERDICT r4:
    the training forward ran the 3x3 stage on a VALU kernel, 0.19 ms per step at 0 % MFMA): wmd_head_fused_fwd +
    wmd_head_shiftsum_fwd (or wmd_head_level_fwd at C = 32) with the extra outputs the backward needs -- the LeakyReLU outputs
    of the stacked 1x1 (`mid`, same layout as _StackedHeadsFn's) and the sigmoid outputs.  Backward = adjoint of the synthesis
    (wmd_idwt_haar_bwd, incl. the clamp of the disparity) followed by _stacked_heads_backward, unchanged.

    inputs: x, yl (previous low-pass or None), x_gate, scale_hf, scale_ll, disp_scale, clamp01, then (w1, b1, w3, b3) of the
    + head, the - head and optionally the LL head.  outputs: yh [B,3,H,W], yl_ll [B,1,H,W] (or empty), out, disp, mid."""

    @staticmethod
    def forward(ctx, x, yl, x_gate, scale_hf, scale_ll, disp_scale, clamp01, *params):
        x = _c(x)
        heads = [params[i:i + 4] for i in range(0, len(params), 4)]       # [+, -, (LL)]
        has_ll = len(heads) == 3
        order = ([2] if has_ll else []) + [0, 1]                            # channel order of mid: [LL, +, -]
        offs, o = {}, 0
        for k in order:
            offs[k] = o
            o += heads[k][0].shape[0]
        B, _, H, W = x.shape
        dev = x.device
        mid = torch.empty((B, o, H, W), device=dev, dtype=torch.float32)
        sp = torch.empty((B, 3, H, W), device=dev, dtype=torch.float32)
        sn = torch.empty_like(sp)
        sl = torch.empty((B, 1, H, W), device=dev, dtype=torch.float32) if has_ll else None
        train = dict(mid=mid, off_p=offs[0], off_n=offs[1], off_ll=offs.get(2, 0), sig_p=sp, sig_n=sn, sig_ll=sl)
        # (the parameters themselves, not detached views: the packed weight images are memoised on the parameter objects)
        res = head_fused_level_nograd(x, tuple(heads[0]), tuple(heads[1]), scale_hf, yl=None if has_ll else _c(yl.detach()),
                                      disp_scale=disp_scale, clamp01=clamp01, head_ll=tuple(heads[2]) if has_ll else None,
                                      scale_ll=scale_ll, train=train)
        yh, out, disp = res[0].squeeze(1), res[1], res[2]
        yl_ll = res[3] if has_ll else yh.new_empty(0)
        ctx.save_for_backward(x, mid, sp, sn, sl, out, *params)
        ctx.meta = (x_gate, float(scale_hf), float(scale_ll), has_ll, order, offs)
        ctx.cfg = (float(disp_scale), bool(clamp01))
        ctx.mark_non_differentiable(mid)
        return yh, yl_ll, out, disp, mid

    @staticmethod
    def backward(ctx, d_yh, d_yl_ll, d_out, d_disp, _d_mid):
        l = _lib.lib()
        x, mid, sp, sn, sl, out = ctx.saved_tensors[:6]
        params = ctx.saved_tensors[6:]
        has_ll = ctx.meta[3]
        disp_scale, clamp01 = ctx.cfg
        B, _, H, W = x.shape
        # adjoint of the synthesis: d(out) + d(disp) through the clamp -> d(low-pass input), d(yh)
        d_lo = torch.empty((B, 1, H, W), device=x.device, dtype=torch.float32)
        d_hi = torch.empty((B, 1, 3, H, W), device=x.device, dtype=torch.float32)
        d_out = _c(d_out) if d_out is not None else None
        d_disp = _c(d_disp) if d_disp is not None else None
        if d_out is None and d_disp is None:
            d_lo.zero_()
            d_hi.zero_()
        else:
            check(l.wmd_idwt_haar_bwd(ptr(d_out), ptr(d_disp), ptr(out), ptr(d_lo), ptr(d_hi), B, H, W, float(disp_scale),
                                      int(clamp01), current_stream()), "wmd_idwt_haar_bwd")
        d_hi = d_hi.view(B, 3, H, W)
        if d_yh is not None:
            d_hi = d_hi + d_yh
        d_ll = None
        if has_ll:      # the low-pass head's output is the synthesis' low-pass input (and an output of its own)
            d_ll = d_lo if (d_yl_ll is None or d_yl_ll.numel() == 0) else d_lo + d_yl_ll
        dx, grads = _stacked_heads_backward(x, mid, sp, sn, sl, params, ctx.meta, d_hi, d_ll, ctx.needs_input_grad[0])
        return (dx, None if has_ll else d_lo, None, None, None, None, None) + tuple(grads)


def fused_level_train(x, head_p, head_n, scale_hf, yl=None, disp_scale=1.0, clamp01=True, head_ll=None, scale_ll=1.0, x_gate=None):
    """Training-mode heads + Haar synthesis of one level on the fused kernels (see _FusedLevelFn; fused_train_supported says
    when).  head_* = (w1, b1, w3, b3); exactly one of yl / head_ll supplies the low-pass input.
    Returns (yh [B,3,H,W], yl_ll [B,1,H,W] or None, out [B,1,2H,2W], disp, mid)."""
    if (yl is None) == (head_ll is None):
        raise _lib.WmdError("fused_level_train: exactly one of yl / head_ll")
    params = tuple(head_p) + tuple(head_n) + (tuple(head_ll) if head_ll is not None else ())
    _require_gpu(x, *params)
    yh, yl_ll, out, disp, mid = _FusedLevelFn.apply(x, yl, x_gate, scale_hf, scale_ll, disp_scale, clamp01, *params)
    return yh, (yl_ll if head_ll is not None else None), out, disp, mid


# ---------------------------------------------------------------------------------------------
# Haar transforms
# ---------------------------------------------------------------------------------------------

class _IdwtFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, yl, yh, disp_scale, clamp01, want_disp):
        l = _lib.lib()
        yl, yh = _c(yl), _c(yh)
        B, Cc, h, w = yl.shape
        out = torch.empty((B, Cc, 2 * h, 2 * w), device=yl.device, dtype=torch.float32)
        disp = torch.empty_like(out) if want_disp else None
        check(l.wmd_idwt_haar_fwd(ptr(yl), ptr(yh), ptr(out), ptr(disp), B * Cc, h, w, float(disp_scale),
                                  int(clamp01), current_stream()), "wmd_idwt_haar_fwd")
        ctx.save_for_backward(out)
        ctx.cfg = (disp_scale, clamp01, want_disp, tuple(yl.shape), tuple(yh.shape))
        if want_disp:
            return out, disp
        return out, out.new_empty(0)

    @staticmethod
    def backward(ctx, d_out, d_disp):
        l = _lib.lib()
        (out,) = ctx.saved_tensors
        disp_scale, clamp01, want_disp, sl, sh = ctx.cfg
        d_yl = torch.empty(sl, device=out.device, dtype=torch.float32)
        d_yh = torch.empty(sh, device=out.device, dtype=torch.float32)
        d_out = _c(d_out) if d_out is not None else None
        d_disp = _c(d_disp) if (want_disp and d_disp is not None) else None
        check(l.wmd_idwt_haar_bwd(ptr(d_out), ptr(d_disp), ptr(out), ptr(d_yl), ptr(d_yh), sl[0] * sl[1], sl[2], sl[3],
                                  float(disp_scale), int(clamp01), current_stream()), "wmd_idwt_haar_bwd")
        return d_yl, d_yh, None, None, None


def idwt_haar(yl, yh, disp_scale=None, clamp01=False):
    """yl [B,C,h,w], yh [B,C,3,h,w] -> (out [B,C,2h,2w], disp or None) with disp = clamp?(out*disp_scale)."""
    _require_gpu(yl, yh)
    if yh.dim() != 5 or yh.shape[2] != 3 or yh.shape[:2] != yl.shape[:2] or yh.shape[3:] != yl.shape[2:]:
        raise _lib.WmdError("idwt_haar: yl %s / yh %s mismatch" % (tuple(yl.shape), tuple(yh.shape)))
    want = disp_scale is not None
    out, disp = _IdwtFn.apply(yl, yh, 1.0 if disp_scale is None else disp_scale, clamp01, want)
    return out, (disp if want else None)


def dwt_haar(x, J=1):
    """J-level Haar analysis of x [B,C,H,W] = DWT(J, "haar", mode="reflect"): returns (yl, [yh_fine..yh_coarse]).  An odd
    axis is extended by one reflected sample (pytorch_wavelets' padding for mode="reflect"), so every level has
    ceil(size/2) coefficients.  The reference only uses it on ground truth (no gradient, NYUv2/train.py:289)."""
    _require_gpu(x)
    l = _lib.lib()
    ll = _c(x.detach())
    yh = []
    for _ in range(J):
        B, Cc, H, W = ll.shape
        h, w = (H + 1) // 2, (W + 1) // 2
        nl = torch.empty((B, Cc, h, w), device=x.device, dtype=torch.float32)
        nh = torch.empty((B, Cc, 3, h, w), device=x.device, dtype=torch.float32)
        if H % 2 or W % 2:
            check(l.wmd_dwt_haar_reflect_fwd(ptr(ll), ptr(nl), ptr(nh), B * Cc, H, W, current_stream()), "wmd_dwt_haar_reflect_fwd")
        else:
            check(l.wmd_dwt_haar_fwd(ptr(ll), ptr(nl), ptr(nh), B * Cc, h, w, current_stream()), "wmd_dwt_haar_fwd")
        yh.append(nh)
        ll = nl
    return ll, yh


# ---------------------------------------------------------------------------------------------
# multi-scale loss front-end
# ---------------------------------------------------------------------------------------------

class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W, align_corners, depth_range):
        l = _lib.lib()
        x = _c(x)
        B, Cc, h, w = x.shape
        y = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
        depth = torch.empty_like(y) if depth_range is not None else None
        lo, hi = depth_range if depth_range is not None else (1.0, 2.0)
        check(l.wmd_upsample_bilinear_fwd(ptr(x), ptr(y), ptr(depth), B * Cc, h, w, H, W, int(align_corners), float(lo),
                                          float(hi), current_stream()), "wmd_upsample_bilinear_fwd")
        ctx.save_for_backward(depth)
        ctx.cfg = (tuple(x.shape), H, W, align_corners, depth_range)
        if depth is None:
            return y, y.new_empty(0)
        return y, depth

    @staticmethod
    def backward(ctx, dy, ddepth):
        l = _lib.lib()
        (depth,) = ctx.saved_tensors
        shp, H, W, align_corners, depth_range = ctx.cfg
        dx = torch.empty(shp, device=dy.device, dtype=torch.float32)
        lo, hi = depth_range if depth_range is not None else (1.0, 2.0)
        use_dd = depth_range is not None and ddepth is not None
        check(l.wmd_upsample_bilinear_bwd(ptr(_c(dy)), ptr(_c(ddepth)) if use_dd else None, ptr(depth) if use_dd else None,
                                          ptr(dx), shp[0] * shp[1], shp[2], shp[3], H, W, int(align_corners), float(lo),
                                          float(hi), current_stream()), "wmd_upsample_bilinear_bwd")
        return dx, None, None, None, None


def upsample_bilinear(x, size, align_corners=False, depth_range=None):
    """F.interpolate(x, size, mode="bilinear", align_corners) -> y; with depth_range=(min_depth, max_depth) also
    returns disp_to_depth(y)[1] (KITTI/trainer.py:337-342).  Differentiable."""
    _require_gpu(x)
    y, depth = _UpsampleFn.apply(x, int(size[0]), int(size[1]), bool(align_corners), depth_range)
    return (y, depth) if depth_range is not None else y
