/*
 * wmd.h — C ABI of libwmd_hip.so: the MI355X (gfx950) implementation of the
 * wavelet-monodepth decoder + Haar IDWT/DWT hot path.
 *
 * The reference (nianticlabs/wavelet-monodepth) has no FFI layer: its boundary
 * for this path is the Python nn.Module surface.  Every entry
 * point below therefore names the reference *operator* it replaces (file:line
 * under /root/reference) — these are what a maintainer binds with ctypes from
 * the decoder modules.
 *
 * Conventions
 *   - all tensors fp32, NCHW, contiguous, device memory owned by the caller
 *   - every call is asynchronous, ordered on `stream` (a hipStream_t passed as void*)
 *   - return value: 0 (WMD_OK) or a negative wmd_status; nothing throws
 *   - the library keeps no state besides a thread-local last-error string
 */
#ifndef WMD_H
#define WMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WMD_VERSION 100 /* 0.1.0 */

typedef enum {
    WMD_OK = 0,
    WMD_ERR_BAD_ARG = -1,     /* null pointer, negative size, unknown enum */
    WMD_ERR_BAD_SHAPE = -2,   /* shape the reference op itself would reject  */
    WMD_ERR_UNSUPPORTED = -3, /* valid but not implemented                   */
    WMD_ERR_HIP = -4,         /* a HIP runtime call or launch failed         */
    WMD_ERR_WORKSPACE = -5,   /* caller-provided workspace too small         */
    WMD_ERR_COMM = -6         /* RCCL failure                                */
} wmd_status;

/* padding applied before a 3x3 convolution
 *   ZERO      nn.ZeroPad2d(1)        KITTI/layers.py:155, NYUv2/networks/layers.py:22
 *   REFLECT   nn.ReflectionPad2d(1)  KITTI/layers.py:153, NYUv2/networks/layers.py:18
 *   REPLICATE nn.ReplicationPad2d(1) NYUv2/networks/layers.py:20                     */
typedef enum { WMD_PAD_ZERO = 0, WMD_PAD_REFLECT = 1, WMD_PAD_REPLICATE = 2 } wmd_pad;

/* activation fused after bias
 *   ELU (alpha=1)  KITTI/layers.py:131      LEAKY(slope) depth_decoder.py:105, NYUv2 layers.py:61
 *   SIGMOID        depth_decoder.py:133     */
typedef enum { WMD_ACT_NONE = 0, WMD_ACT_ELU = 1, WMD_ACT_LEAKY = 2, WMD_ACT_SIGMOID = 3 } wmd_act;

int wmd_version(void);
const char* wmd_last_error(void);
const char* wmd_status_string(int status);

/* Opt-in kernel timing (the only process-global state in the library; not thread-safe).
 * Between begin and end every kernel launch of this library is bracketed by a hipEvent pair on
 * its launch stream.  wmd_profile_end synchronises and writes a JSON array
 *   [{"kernel": name, "calls": n, "ms": total, "flops": algorithmic FLOPs, "bytes": algorithmic bytes,
 *     "mfma_flops": FLOPs the matrix pipe executes (Winograd kernels: fewer than algorithmic)}, ...]
 * into buf (truncated to cap); returns the number of bytes needed, or a negative wmd_status.    */
int wmd_profile_begin(void);
long wmd_profile_end(char* buf, size_t cap);

/* ------------------------------------------------------------------ *
 * Haar wavelets  (third-party pytorch_wavelets; call sites
 * KITTI/networks/decoders/depth_decoder.py:85,164,372,416,
 * NYUv2/networks/decoders/densedepth_decoder.py:99,129,137,145, NYUv2/train.py:258,289;
 * closed form restated by the authors at depth_decoder.py:225-239)
 * ------------------------------------------------------------------ */

/* IDWT(wave="haar", mode="zero"), one level.
 *   yl [N,h,w], yh [N,3,h,w] (LH,HL,HH)  ->  out [N,2h,2w]
 *   if disp != NULL also writes disp = out*disp_scale, clamped to [0,1] when clamp01 != 0
 *   (depth_decoder.py:166 `clamp(yl / 2**(i-1), 0, 1)`; NYUv2 decoder: scale only).
 *   N = batch*channels (channels is 1 everywhere in the reference).                   */
int wmd_idwt_haar_fwd(const float* yl, const float* yh, float* out, float* disp,
                      int N, int h, int w, float disp_scale, int clamp01, void* stream);

/* Adjoint of the above.  d_out [N,2h,2w] (may be NULL = zero), d_disp [N,2h,2w] (may be NULL);
 * `out` is the forward result (needed for the clamp mask when d_disp != NULL && clamp01).
 * Writes d_yl [N,h,w] and d_yh [N,3,h,w].                                              */
int wmd_idwt_haar_bwd(const float* d_out, const float* d_disp, const float* out,
                      float* d_yl, float* d_yh, int N, int h, int w,
                      float disp_scale, int clamp01, void* stream);

/* DWT(J=1, wave="haar", mode="reflect"/"zero") on even H,W (the only case the reference hits,
 * NYUv2/train.py:258 on 240x320 depth): x [N,2h,2w] -> yl [N,h,w], yh [N,3,h,w].
 * Call J times for a J-level transform (yh list is fine -> coarse).                    */
int wmd_dwt_haar_fwd(const float* x, float* yl, float* yh, int N, int h, int w, void* stream);
/* The same analysis step for ANY input size H x W (mode="reflect" of pytorch_wavelets' DWTForward, NYUv2/train.py:258):
 * an odd axis is extended by one reflected sample on the right / bottom (x[N] = x[N-2]) before the stride-2 filter
 * pair, so yl [N,ceil(H/2),ceil(W/2)], yh [N,3,ceil(H/2),ceil(W/2)].  An odd axis needs >= 2 samples.                */
int wmd_dwt_haar_reflect_fwd(const float* x, float* yl, float* yh, int N, int H, int W, void* stream);

/* ------------------------------------------------------------------ *
 * Dense convolutions of the decoders (implicit GEMM on fp32 MFMA)
 * ------------------------------------------------------------------ */

/* Number of floats of the MFMA-fragment-ordered weight image for a [Cout,Cin,k,k] filter. */
size_t wmd_conv_packed_weight_floats(int Cout, int Cin, int ksize);

/* Re-order nn.Conv2d weights [Cout,Cin,k,k] (k = 1 or 3) into the fragment image `wp`
 * the conv kernels read (16 out-channels x 4 in-channels per 256-byte wave load).       */
int wmd_conv_pack_weights(const float* w, float* wp, int Cout, int Cin, int ksize, void* stream);

/* Same re-ordering for the data-gradient pass: transposes in/out channels and flips taps. */
int wmd_conv_pack_weights_dgrad(const float* w, float* wp, int Cout, int Cin, int ksize, void* stream);

typedef struct {
    int B, H, W;        /* output size; also the logical input size (stride 1, "same")  */
    int C1;             /* channels of x1                                              */
    int up1;            /* 1: x1 is [B,C1,H,W]; 2: x1 is [B,C1,H/2,W/2], nearest x2    */
    int C2;             /* channels of the skip tensor x2 [B,C2,H,W]; 0 if none        */
    int Cout;
    int ksize;          /* 1 or 3                                                      */
    int pad_mode;       /* wmd_pad (ignored for ksize 1)                               */
    int act;            /* wmd_act                                                     */
    float slope;        /* LeakyReLU negative slope                                    */
    const float* x1;
    const float* x2;    /* may be NULL when C2 == 0                                    */
    const float* wp;    /* packed weights (wmd_conv_pack_weights) for Cin = C1 + C2    */
    const float* bias;  /* [Cout] or NULL                                              */
    float* y;           /* [B,Cout,H,W]                                                */
    float* workspace;   /* split-K partial sums; may be NULL -> never split            */
    size_t workspace_floats;
    int tune_cfg;       /* 0: built-in cost model; k>0: force tile configuration k-1     */
    int tune_ksplit;    /* 0: cost model; k>0: force k-way split of the Cin reduction,
                           finished inside the convolution where the kernel can (the
                           32x32x2 Winograd kernels: the last K-slice block of a tile sums
                           the slices -- same order, same bits as the second-stage kernel);
                           k<0: |k| slices summed by the second-stage kernel (only for
                           kernels that have both forms; WMD_ERR_UNSUPPORTED otherwise)     */
    const float* wp_wino; /* optional (3x3 only): wmd_conv_pack_weights_wino image; enables the Winograd F(2x2,3x3)
                             configurations ("conv_wino_kernel<...>": 2.25x fewer MFMAs, ~1e-6 relative rounding) */
    const float* gate;  /* optional [B,Cout,H,W]: y = act(conv + bias) * act_gate'(gate), the derivative written in terms of
                           an activation OUTPUT (ELU: g > 0 ? 1 : g + 1; LeakyReLU: g > 0 ? 1 : slope).  Lets a consumer's
                           data gradient hand the producer dz = dy * f'(y) directly (no separate wmd_act_bwd pass).
                           Direct kernels only (the Winograd configurations are skipped when it is set).             */
    int gate_act;       /* wmd_act of the gate                                          */
    float gate_slope;
    /* Block-sparse execution (3x3; the threshold-gated sparse decoders on the dense MFMA kernels when several frames are
     * decoded together): with out_mask [B,H,W] only the pixel tiles that contain an active output pixel are computed (every
     * block tests its own tile's mask bytes and returns when there is none), outputs outside the mask are written as 0 and
     * tiles without active pixels are NOT touched (y must be zero-initialised -- or every consumer of y must select on the
     * mask rather than multiply by it, which is how the sparse decoders use their never-refilled activation pool: the next
     * layer's in_mask / the head's yh_mask gate every read, tests/test_gpu_configs.py poisons the pool with NaN); with in_mask
     * [B,H,W] a padded input
     * position outside the mask reads 0 for x1 and x2 alike -- the mask test after the coordinate padding of
     * sparse_conv3x3 (KITTI/layers.py:439-453).  Split-K stays available (the reduce pass writes 0 outside out_mask and
     * never reads the slots of a skipped tile); implemented by the Winograd kernels (needs wp_wino).
     * in_mask_2x2 != 0 is the caller's promise that in_mask is constant on aligned 2x2 pixel blocks (it is the nearest
     * upsampling of a half-resolution mask, as the decoders' upsample mask is: MaxPool5(upsample(m)) = upsample(MaxPool3(m)));
     * with up1 = 2 it lets the upsampled operand keep the structured low-resolution path of conv_wino32_kernel (the mask
     * of source pixel (y, x) is read at (2y, 2x)).  Without the promise a masked upsampled layer runs the generic gather.  */
    const uint8_t* in_mask;
    const uint8_t* out_mask;
    int in_mask_2x2;
    /* Work-list form of the block-sparse execution (round 4; wmd_mask_level_lists builds the list).  out_tiles holds the
     * indices ((b * tiles_y + ty) * tiles_x + tx, tiles of out_tile_h x out_tile_w pixels) of the pixel tiles that contain
     * an active pixel of out_mask, per frame: out_tiles [B][tiles_y * tiles_x], out_tile_count [B] (frame f's tiles are the
     * first out_tile_count[f] entries of its segment, in any order).  Only listed tiles are
     * dispatched to a matrix-pipe workgroup -- an unlisted tile costs nothing and, as before, is not written -- and the
     * split of the input-channel reduction is chosen ON THE DEVICE from the count (few active tiles: every tile's K loop is
     * spread over many workgroups, whose partial sums meet in a list-driven second pass; many: no split, no second pass
     * work), so that the time follows the active work.  Needs out_mask, a 3x3 layer, wp_wino, a `pure` channel layout
     * (C1, C2 multiples of 8; an upsampled x1 needs in_mask_2x2) and a tile shape the library has a list kernel for
     * (wmd_conv_list_tile_supported); workspace: wmd_conv_fwd_workspace_floats.  Values: identical to the mask form.     */
    const int32_t* out_tiles;
    const int32_t* out_tile_count;
    int out_tile_h, out_tile_w;
} wmd_conv_args;

/* != 0 when wmd_conv_fwd has a work-list kernel for pixel tiles of tile_h x tile_w */
int wmd_conv_list_tile_supported(int tile_h, int tile_w);

/* Fused  upsample(x1) ++ x2  ->  pad  ->  conv kxk  ->  + bias  ->  activation.
 * Replaces ConvBlock/Conv3x3/Conv1x1 (KITTI/layers.py:120-173), `upsample` (:233-236),
 * the skip `torch.cat` (depth_decoder.py:146-149), NYUv2 Conv3x3/UpSampleBlock
 * (NYUv2/networks/layers.py:11-32,57-67).                                              */
int wmd_conv_fwd(const wmd_conv_args* args, void* stream);

/* Suggested workspace size (floats) for wmd_conv_fwd on this problem. */
size_t wmd_conv_fwd_workspace_floats(const wmd_conv_args* args);

/* Winograd-domain weight image of a 3x3 filter: U = G g G^T per (out, in) channel pair, in the fragment order of
 * wmd_conv_pack_weights with 16 transformed positions in place of the 9 taps.  dgrad != 0: the transposed / flipped
 * filter of the data-gradient pass.                                                                              */
size_t wmd_conv_packed_weight_floats_wino(int Cout, int Cin);
int wmd_conv_pack_weights_wino(const float* w, float* wp, int Cout, int Cin, int dgrad, void* stream);

/* Every weight image of a set of filters in ONE launch (training repacks all of them after each optimizer step).  Each
 * non-NULL output receives exactly what the single-filter entry point writes: fwd = wmd_conv_pack_weights, dgrad =
 * wmd_conv_pack_weights_dgrad, wino_fwd / wino_dgrad = wmd_conv_pack_weights_wino(dgrad = 0 / 1) (3x3 filters only).
 * Outputs may alias slices of one buffer (sizes: wmd_conv_packed_weight_floats[_wino]).                              */
typedef struct {
    const float* w;      /* [Cout,Cin,ksize,ksize]                                       */
    int Cout, Cin, ksize;
    float* fwd;
    float* dgrad;
    float* wino_fwd;
    float* wino_dgrad;
} wmd_pack_item;
int wmd_conv_pack_many(const wmd_pack_item* items, int n, void* stream);

/* Tile-configuration table (for callers that autotune: wavelet_monodepth_amd/tuner.py).
 * wmd_conv_config_name returns e.g. "conv_fwd_kernel<8,32,2,4,1,4,8,9>" (the kernel's template
 * arguments TH,TW,MR,NR,WM,WN,CK,TAPS) or NULL when i is out of range.                        */
int wmd_conv_num_configs(void);
const char* wmd_conv_config_name(int i);

/* dz = dy * act'(y)  (y = the activation OUTPUT saved by the forward). In place allowed. */
int wmd_act_bwd(const float* dy, const float* y, float* dz, size_t n, int act, float slope, void* stream);

typedef struct {
    int B, H, W;        /* size of dz / of the logical (upsampled, concatenated) input   */
    int C1, up1, C2, Cout, ksize, pad_mode;
    const float* dz;    /* [B,Cout,H,W] gradient w.r.t. the pre-activation              */
    const float* wp_dgrad; /* wmd_conv_pack_weights_dgrad image                         */
    float* dx1;         /* [B,C1,H/up1,W/up1] or NULL                                   */
    float* dx2;         /* [B,C2,H,W] or NULL                                           */
    float* workspace;   /* needs wmd_conv_dgrad_workspace_floats                        */
    size_t workspace_floats;
    int tune_cfg;       /* as in wmd_conv_args: 0 = cost model, k > 0 forces configuration k-1 */
    int tune_ksplit;
    const float* wp_dgrad_wino; /* optional (3x3): wmd_conv_pack_weights_wino(..., dgrad = 1) image, enables Winograd */
    const float* x1_fwd; /* optional: the forward's x1 [B,C1,H/up1,W/up1], itself the OUTPUT of an activation x1_act: dx1 is
                            then multiplied by x1_act'(x1), i.e. it is already the producer's pre-activation gradient     */
    int x1_act;          /* wmd_act (WMD_ACT_NONE: no gating)                            */
    float x1_slope;
} wmd_conv_dgrad_args;

size_t wmd_conv_dgrad_workspace_floats(const wmd_conv_dgrad_args* args);
/* Data gradient: adjoint of pad + concat + nearest-upsample folded behind a full correlation. */
int wmd_conv_dgrad(const wmd_conv_dgrad_args* args, void* stream);

typedef struct {
    int B, H, W;
    int C1, up1, C2, Cout, ksize, pad_mode;
    const float* x1;
    const float* x2;
    const float* dz;    /* [B,Cout,H,W]                                                 */
    float* dw;          /* [Cout,C1+C2,k,k]  (overwritten)                              */
    float* dbias;       /* [Cout] or NULL    (overwritten)                              */
    float* workspace;
    size_t workspace_floats;
    int tune_cfg;       /* 0 = library model; k > 0 forces entry k-1 of the Winograd F(2x2,3x3) weight-gradient table
                           (wmd_conv_wgrad_num_configs / _config_name; 3x3 only); -1 = direct kernel, library's tile */
    int tune_nsplit;    /* 0 = library model; > 0 forces the number of pixel-tile slices (= partial sums)          */
} wmd_conv_wgrad_args;

/* the Winograd weight-gradient tile table (for autotuners; names are stable across versions) */
int wmd_conv_wgrad_num_configs(void);
const char* wmd_conv_wgrad_config_name(int index);

size_t wmd_conv_wgrad_workspace_floats(const wmd_conv_wgrad_args* args);
/* Weight + bias gradient (deterministic two-stage split over the B*H*W reduction). */
int wmd_conv_wgrad(const wmd_conv_wgrad_args* args, void* stream);

/* Depthwise 3x3 + ReLU over the same virtual input (upsample(x1) ++ x2, padded): the first half of the NYUv2 decoders'
 * optional `is_depthwise` Conv3x3 (NYUv2/networks/layers.py:23-25,70-79); its bias-free 1x1 second half is an
 * ordinary wmd_conv_fwd.  w [C1+C2,1,3,3]; y, dy [B,C1+C2,H,W].  Backward: dx1 / dx2 / dw may each be NULL.          */
typedef struct {
    int B, H, W;
    int C1, up1, C2, pad_mode;
    const float* x1;
    const float* x2;
    const float* w;
} wmd_dwconv_args;
int wmd_dwconv3x3_fwd(const wmd_dwconv_args* args, float* y, void* stream);
size_t wmd_dwconv3x3_bwd_workspace_floats(const wmd_dwconv_args* args);
int wmd_dwconv3x3_bwd(const wmd_dwconv_args* args, const float* y, const float* dy, float* dx1, float* dx2, float* dw,
                      float* workspace, size_t workspace_floats, void* stream);

/* ------------------------------------------------------------------ *
 * Wavelet heads: 3x3 convolutions with 1..3 output channels (HBM-bound)
 * ------------------------------------------------------------------ */
typedef struct {
    int B, H, W, C;     /* input [B,C,H,W]                                              */
    int Cout;           /* 1..4                                                         */
    int pad_mode;
    int mode;           /* 0: y = scale*conv_p(xp)                     (NYUv2 wave*, :122-141)
                           1: y = scale*sigmoid(conv_p(xp))            (LL head, depth_decoder.py:133)
                           2: y = scale*(sigmoid(conv_p(xp)) - sigmoid(conv_n(xn)))  (:134-135) */
    float scale;
    const float* xp; const float* wgt_p; const float* bias_p;   /* wgt [Cout,C,3,3]     */
    const float* xn; const float* wgt_n; const float* bias_n;   /* mode 2 only          */
    float* y;           /* [B,Cout,H,W]                                                 */
    float* sig_p;       /* optional [B,Cout,H,W]: sigmoid outputs saved for backward    */
    float* sig_n;
    size_t xp_bstride;  /* floats between consecutive batch items of xp / xn; 0 = C*H*W (lets xp, xn be  */
    size_t xn_bstride;  /* channel slices of one stacked [B,Ctot,H,W] tensor)                             */
    float* workspace;   /* optional (wmd_head3x3_workspace_floats): lets coarse levels split the channel  */
    size_t workspace_floats; /* loop over several workgroups (two-stage, deterministic)                   */
} wmd_head_args;

/* Replaces Conv3x3(C,3|1) + Sigmoid + the 2^(s-1)(sigma+ - sigma-) combine of
 * DepthWaveProgressiveDecoder.get_coefficients (depth_decoder.py:126-136) and the
 * NYUv2 wave1_ll/wave{1,2,3} convolutions (densedepth_decoder.py:106-115).             */
int wmd_head3x3_fwd(const wmd_head_args* args, void* stream);
size_t wmd_head3x3_workspace_floats(const wmd_head_args* args);

/* Fused inference form of a level's two high-frequency heads (depth_decoder.py:108-136):
 *   mid_s = LeakyReLU(W1_s x + b1_s)            s in {+,-}, both C -> C     (MFMA GEMM, result stays on chip: for C = 64,
 *                                                                            128, 256 in the accumulator registers, which are
 *                                                                            the next product's operand -- head_chain_kernel)
 *   t[s*27 + co*9 + tap] = sum_c W3_s[co,c,tap] * mid_s[c]                  (second MFMA GEMM of the same wave)
 * i.e. the 3x3 convolution regrouped as 27 tap-partial 1x1 outputs; wmd_head_shiftsum_fwd then gathers the
 * nine shifted taps, adds the bias, applies sigmoid / 2^(s-1)(sig+ - sig-) and (optionally) the Haar IDWT.
 * wp1: wmd_conv_pack_weights image of the stacked [2C, C, 1, 1] filter (+ rows first); bias1 [2C];
 * wp2: two wmd_conv_pack_weights images of [27, C, 1, 1] (W3.permute(0,2,3,1).reshape(27, C)), + then -.
 * C in {32, 64, 128, 256}; other widths return WMD_ERR_UNSUPPORTED (callers use the unfused operators).       */
typedef struct {
    int B, H, W, C;
    float slope;
    const float* x;      /* [B,C,H,W]      */
    const float* wp1;
    const float* bias1;
    const float* wp2;
    float* t;            /* [B,t_planes,H,W]  */
    int chain;           /* 0: the + and - chains -> t planes 0..53.
                            1: the coarsest level's low-pass chain (C -> C/4 -> 1, depth_decoder.py:104-106,126-129) as a
                               second launch over the same x: wp1 / bias1 = its [C/4, C, 1, 1] filter, wp2 = ONE
                               [27, C/4, 1, 1] image whose rows 0..8 hold the nine taps -> t planes 54..62 of an 81-plane t,
                               completed by wmd_head_shiftsum_fwd(yl_out).  C = 256 only (WMD_ERR_UNSUPPORTED otherwise)   */
    int t_planes;        /* planes per image of t: 0 or 54, or 81 when the low-pass chain shares the buffer              */
    /* optional [B,H,W] bytes (round 4, chain = 0 on the chained kernel): a run of pixels without a set byte is skipped (its
     * t planes are not written).  The sparse decoders pass the upconv1 mask = the 3x3 dilation of the wavelet mask: exactly
     * the pixels whose tap-partials a surviving output of wmd_head_shiftsum_fwd(yh_mask) gathers.                          */
    const uint8_t* run_mask;
    /* optional (round 4, chain = 0, C = 256, t_planes = 81): the low-pass chain's operands (what a chain = 1 call would take as
     * wp1 / bias1 / wp2).  The same call then fills planes 54..62 too -- on the chained kernel as a third group of workgroups
     * of the SAME launch (a quarter of a side's first product: an LL workgroup walks four pixel tiles, so the launch stays
     * inside one round of workgroup slots), otherwise as a second launch issued by the library.                            */
    const float* ll_wp1;
    const float* ll_bias1;
    const float* ll_wp2;
    /* optional (round 5, training forward on the fused kernels: trainer.py:208-212 needs what autograd would have kept): the
     * LeakyReLU outputs of the 1x1 stage are ALSO written to mid_out [B, mid_ct, H, W], channels mid_off_p.. (+ side),
     * mid_off_n.. (- side) and, with the low-pass chain riding along, mid_off_ll.. (C/4 channels) -- the layout the heads'
     * backward operators (wmd_head3x3_bwd / wmd_head1x1_bwd) read.  Chained kernel only (C = 64, 128, 256, chain = 0):
     * WMD_ERR_UNSUPPORTED otherwise.                                                                                      */
    float* mid_out;
    int mid_ct, mid_off_p, mid_off_n, mid_off_ll;
} wmd_head_fused_args;
int wmd_head_fused_fwd(const wmd_head_fused_args* args, void* stream);
/* Round 6: the first stage of up to three levels in ONE launch (chained widths 64 / 128 / 256, no run_mask / mid_out; anything else
 * falls back to n_levels calls of wmd_head_fused_fwd).  The heads of a level read only that level's trunk activation
 * (depth_decoder.py:126-136,164-166), so a dense decoder may postpone levels 4..2 until `upconv(2,1)` is done: alone none of them
 * fills 256 CUs, together they balance.  Same tap-partial planes, bit for bit.                                                  */
int wmd_head_fused_multi_fwd(const wmd_head_fused_args* levels, int n_levels, void* stream);

typedef struct {
    int B, H, W;
    int pad_mode;        /* padding of the 3x3 it completes (reflect for the KITTI heads)                    */
    float scale;         /* 2^(s-1)                                                                          */
    const float* t;      /* [B,54,H,W] from wmd_head_fused_fwd                                               */
    const float* bias_p; /* [3] */
    const float* bias_n; /* [3] */
    float* yh;           /* [B,3,H,W] = scale*sigmoid(.) - scale*sigmoid(.)                                  */
    /* optional fused IDWT (wmd_idwt_haar_fwd semantics): yl [B,H,W] -> out [B,2H,2W], disp                  */
    const float* yl;
    float* out;
    float* disp;
    float disp_scale;
    int clamp01;
    /* optional low-pass head completed in the same pass (t then has 81 planes per image, see wmd_head_fused_args.chain):
     * yl_out [B,H,W] = scale_ll * sigmoid(bias_ll[0] + nine shifted taps of planes 54..62); it is also the low-pass input
     * of the synthesis (yl must be NULL).                                                                                */
    const float* bias_ll; /* [1] or NULL */
    float scale_ll;
    float* yl_out;
    /* optional [B,H,W] bytes: yh is zeroed where the mask is 0 before it is stored and synthesised (the wavelet mask of the
     * sparse decoders' dense branch, depth_decoder.py:272)                                                              */
    const uint8_t* yh_mask;
    /* optional [B][2] uint32 (round 4): the order-preserving keys of the running (min, max) of every frame's `out` (the new
     * low-pass plane) are folded in with one atomic pair per wavefront -- the range the NEXT level's threshold needs
     * (depth_decoder.py:308: yl.max() - yl.min()) without a reduction pass of its own.  key(f) = bits(f) ^ (f < 0 ?
     * 0xFFFFFFFF : 0x80000000); armed state (0xFFFFFFFF, 0); consumed by wmd_mask_level_lists.                             */
    uint32_t* range_keys;
    /* optional (round 5, training forward): the sigmoid outputs themselves -- sig_p, sig_n [B,3,H,W], sig_ll [B,1,H,W] (with
     * yl_out) -- whose s (1 - s) the backward of the heads multiplies by (what wmd_head3x3_fwd returns as sig_p / sig_n)      */
    float* sig_p;
    float* sig_n;
    float* sig_ll;
} wmd_head_shiftsum_args;
int wmd_head_shiftsum_fwd(const wmd_head_shiftsum_args* args, void* stream);
/* Round 5: the completions of up to three CONSECUTIVE levels (coarse to fine, each twice the size of the one before) in one launch --
 * level k's synthesis output is level k+1's low-pass input (depth_decoder.py:164: yl = IDWT((yl, [yh])) feeds the next iteration), so
 * levels[k > 0] carry yl = yl_out = NULL and take it from the chain; every level needs `out`.  Dense inference only (no yh_mask,
 * range_keys or sigmoid outputs).  Same values as n_levels calls of wmd_head_shiftsum_fwd.                                      */
int wmd_head_shiftsum_chain_fwd(const wmd_head_shiftsum_args* levels, int n_levels, void* stream);

/* Single-launch inference form of a level's two high-frequency heads AND the Haar synthesis that consumes them
 * (depth_decoder.py:108-136,164-166): 1x1 -> LeakyReLU -> 3x3 (as tap-partials) -> sigmoid -> 2^(s-1)(sig+ - sig-)
 * -> IDWT -> clamp, with every intermediate in LDS (wmd_head_level.hip).  Same operands as wmd_head_fused_fwd +
 * wmd_head_shiftsum_fwd, which remain the route for the widths this kernel does not cover.
 * wmd_head_level_supported(C) != 0 for C in {32, 64}.                                                            */
typedef struct {
    int B, H, W, C;
    int pad_mode;        /* padding of the 3x3 (reflect for the KITTI heads)                                  */
    float slope;         /* LeakyReLU slope of the 1x1 (0.1)                                                  */
    float scale;         /* 2^(s-1)                                                                           */
    const float* x;      /* [B,C,H,W]                                                                         */
    const float* wp1;    /* packed stacked [2C,C,1,1] filter, + rows first (wmd_conv_pack_weights)            */
    const float* bias1;  /* [2C] or NULL                                                                      */
    const float* wp2;    /* two packed [27,C,1,1] images (row co*9+tap), + then -                             */
    const float* bias_p; /* [3] or NULL                                                                       */
    const float* bias_n; /* [3] or NULL                                                                       */
    float* yh;           /* [B,3,H,W]                                                                         */
    const float* yl;     /* optional [B,H,W]: fused wmd_idwt_haar_fwd                                         */
    float* out;          /* [B,2H,2W] (with yl)                                                               */
    float* disp;         /* optional [B,2H,2W] = clamp(out * disp_scale)                                      */
    float disp_scale;
    int clamp01;
    const uint8_t* yh_mask; /* optional [B,H,W] bytes, see wmd_head_shiftsum_args.yh_mask                    */
    /* optional (round 5, training forward): the LeakyReLU outputs of the 1x1 stage -> mid_out [B, mid_ct, H, W] at channels
     * mid_off_p.. / mid_off_n.. (see wmd_head_fused_args.mid_out) and the sigmoid outputs sig_p, sig_n [B,3,H,W]             */
    float* mid_out;
    int mid_ct, mid_off_p, mid_off_n;
    float* sig_p;
    float* sig_n;
} wmd_head_level_args;
int wmd_head_level_supported(int C);
int wmd_head_level_fwd(const wmd_head_level_args* args, void* stream);
/* Round 6: the level's heads + synthesis AND the completions of up to three coarser levels in ONE launch (the streaming kernel's
 * pyramid, wmd_head_stream.hip): coarse[k] as for wmd_head_shiftsum_chain_fwd (coarse to fine; the finest is half this level's size
 * and its synthesis output IS this level's low-pass input, depth_decoder.py:164: args->yl must be NULL).  Same values, bit for bit,
 * as wmd_head_shiftsum_chain_fwd followed by wmd_head_level_fwd.  Dense inference outputs only; C = 32, H and W multiples of 8.    */
int wmd_head_level_pyramid_supported(int C, int B, int H, int W);   /* 0: no; 1: runs; 2: runs and is expected to pay (enough CUs for its <= 64-row segments) */
int wmd_head_level_pyramid_fwd(const wmd_head_level_args* args, const wmd_head_shiftsum_args* coarse, int n_coarse, void* stream);

/* Backward of the 3x3 stage of a level's wavelet heads (training): the 2-3 heads Conv3x3(C, 3 | 1) + sigmoid of a level
 * (depth_decoder.py:104-136; backward from torch.autograd, KITTI/trainer.py:211) given the pre-sigmoid gradients dy3
 * [B,n_out,H,W] (rows [LL], +, -) and the LeakyReLU outputs mid [B,Ct,H,W] of the stacked 1x1 stage:
 *   dzmid = act'(mid) * conv3x3^T(dy3)  (the 1x1 stage's pre-activation gradient),  dw3 / db3 per head.
 * One pass each over mid for the data and the weight gradient, no padded-domain buffer (wmd_head_bwd.hip).  At most 24
 * 64-channel slices over all heads (WMD_ERR_UNSUPPORTED beyond: use wmd_conv_dgrad / wmd_conv_wgrad on the stacked filter). */
typedef struct {
    int row0;          /* first plane of dy3 that belongs to this head                   */
    int nrows;         /* its output channels: 3 (high-frequency heads) or 1 (low-pass)  */
    int ch0;           /* first channel of mid that belongs to this head                 */
    int nch;           /* its channel count                                              */
    const float* w3;   /* [nrows, nch, 3, 3]                                             */
    float* dw3;        /* [nrows, nch, 3, 3]                                             */
    float* db3;        /* [nrows]                                                        */
} wmd_head_bwd_head;
typedef struct {
    int B, H, W;
    int Ct;            /* channels of mid                                                */
    int n_out;         /* planes of dy3                                                  */
    int pad_mode;      /* padding of the 3x3 (reflect for the KITTI heads)               */
    int act;           /* activation that produced mid: WMD_ACT_LEAKY (or NONE / ELU)    */
    float slope;
    const float* dy3;  /* [B,n_out,H,W]                                                  */
    const float* mid;  /* [B,Ct,H,W]                                                     */
    float* dzmid;      /* [B,Ct,H,W]; channels of no head are not written                */
    int n_heads;       /* 1..3                                                           */
    wmd_head_bwd_head head[3];
    float* workspace;  /* wmd_head3x3_bwd_workspace_floats(args) floats                  */
    size_t workspace_floats;
} wmd_head3x3_bwd_args;
size_t wmd_head3x3_bwd_workspace_floats(const wmd_head3x3_bwd_args* args);
int wmd_head3x3_bwd(const wmd_head3x3_bwd_args* args, void* stream);

/* Backward of the 1x1 stage of a level's wavelet heads (training): the first layers Conv1x1(C, C | C/4) of every head of a
 * level, stacked along the output channels (depth_decoder.py:104-136), given the pre-activation gradient dz [B,Ct,H,W] of the
 * stack (wmd_head3x3_bwd's dzmid):  dx = act_x'(x) * w1^T dz (optional),  dw1 = sum_p dz (x) x,  db1 = sum_p dz.
 * One pass over dz and x each for the data and the weight gradient (wmd_head_bwd1.hip).                                   */
typedef struct {
    int B, H, W;
    int C;             /* channels of x                                                  */
    int Ct;            /* channels of dz = rows of w1 (a multiple of 8)                  */
    int x_act;         /* activation that produced x (WMD_ACT_ELU for the decoders' trunk; NONE: no gate) */
    float x_slope;
    const float* dz;   /* [B,Ct,H,W]                                                     */
    const float* x;    /* [B,C,H,W]                                                      */
    const float* w1;   /* [Ct,C] (the [Ct,C,1,1] filters of the heads, concatenated)     */
    float* dx;         /* [B,C,H,W] or NULL                                              */
    float* dw1;        /* [Ct,C]                                                         */
    float* db1;        /* [Ct]                                                           */
    float* workspace;  /* wmd_head1x1_bwd_workspace_floats(args) floats                  */
    size_t workspace_floats;
} wmd_head1x1_bwd_args;
size_t wmd_head1x1_bwd_workspace_floats(const wmd_head1x1_bwd_args* args);
int wmd_head1x1_bwd(const wmd_head1x1_bwd_args* args, void* stream);
/* Both stages of a level's high-frequency heads in three launches instead of six: the 3x3 data gradient (-> dzmid); then the
 * 3x3 weight gradient, the 1x1 data gradient and the 1x1 weight gradient -- independent of each other once dzmid exists, each a
 * latency chain on its own -- as ONE launch; then both reduces as one.  Same arguments and results as wmd_head3x3_bwd(a3)
 * followed by wmd_head1x1_bwd(a1) with a1->dz == a3->dzmid; 3-channel heads only (a level with the low-pass head: the two
 * separate calls).  Round 6: a level of two 32-channel heads over a 32-channel x (Ct = 64, whole 64-pixel tiles, W % 4 == 0,
 * dx wanted) runs all four GEMMs as ONE launch (head_bwd_fused32_kernel) + the reduce; dzmid then stays in LDS and the buffer
 * is NOT written (it is scratch of this call either way).  WMD_HEAD_BWD_FUSED=0: always the three-launch form.               */
int wmd_head_bwd(const wmd_head3x3_bwd_args* a3, const wmd_head1x1_bwd_args* a1, void* stream);

/* ------------------------------------------------------------------ *
 * Sparse (threshold-gated) decoder path, batch 1
 *
 * Layout decision: activations stay DENSE [C,H,W] (zero-initialised by the caller); sparsity lives in
 *   - uint8 masks [H,W] (the reference's lowres/upconv0/upsample/upconv1/wavelet masks), and
 *   - compacted raster-order lists of active pixel indices (y*W+x) with a device-side count,
 * so no kernel needs a host-visible nnz and no index map is ever materialised.  The reference's
 * compact-tensor semantics are kept exactly: a neighbour that is not in the input mask reads 0
 * (KITTI/layers.py:439-453), index-map padding becomes coordinate padding followed by the mask test.
 * ------------------------------------------------------------------ */

/* out2[0] = min(x), out2[1] = max(x)   (depth_decoder.py:308 `yl.max() - yl.min()`), n >= 1. */
int wmd_minmax(const float* x, size_t n, float* out2, void* stream);

/* mask[p] = (max_b |yh[b,p]| > (minmax[1]-minmax[0]) * thresh_ratio) ? 1 : 0   (depth_decoder.py:308-309).
 * yh [3,h,w] planes.  The comparison is done exactly as the reference does it in fp32.            */
int wmd_mask_threshold(const float* yh, const float* minmax, float thresh_ratio, uint8_t* mask,
                       int h, int w, void* stream);

typedef struct {
    int up;          /* 1 or 2: nearest-upsample the input mask first (depth_decoder.py:311)       */
    int radius;      /* 0,1,2 ...: MaxPool2d(2r+1, stride 1, padding r) (:313-319)                 */
    uint8_t* out;    /* [h*up, w*up]                                                               */
    int32_t* nnz;    /* optional: the number of set pixels of frame f is ADDED to nnz[f * nnz_stride] (the caller
                        zero-initialises it) -- the pixel count of a mask without a compaction pass               */
    int nnz_stride;
} wmd_dilate_spec;
/* All dilated variants of one mask in a single launch (n <= 8). */
int wmd_mask_dilate_multi(const uint8_t* mask, int h, int w, const wmd_dilate_spec* specs, int n, void* stream);
/* Batched forms (extension: the reference's sparse decoder asserts batch 1, depth_decoder.py:297; on a 256-CU GPU one
 * 640x192 frame cannot fill the machine, so B frames -- each with its own range, threshold mask, pixel lists and counts --
 * go through the SAME launches): mask [B,h,w], every specs[i].out [B,h*up,w*up].                                        */
int wmd_mask_dilate_multi_b(const uint8_t* mask, int B, int h, int w, const wmd_dilate_spec* specs, int n, void* stream);

/* wmd_minmax + wmd_mask_threshold + wmd_mask_dilate_multi of one decoder level in ONE launch
 * (depth_decoder.py:308-319): thr = (max(yl) - min(yl)) * thresh_ratio over yl[n_yl]; base[p] = max_b |yh[b,p]| > thr
 * on the [h,w] grid of yh [3,h,w]; specs[i].out = dilation of the (nearest-upsampled) base mask exactly as
 * wmd_mask_dilate_multi would produce it; a spec (up 1, radius 0) yields the base mask itself.  Bit-identical to the
 * three separate calls (min/max are order-independent, the threshold is the same fp32 expression).          */
int wmd_mask_level(const float* yl, size_t n_yl, const float* yh, float thresh_ratio, int h, int w,
                   const wmd_dilate_spec* specs, int n, void* stream);
/* batched: yl [B,n_yl], yh [B,3,h,w], outputs [B,...]; the range (max - min) is taken per frame.  minmax_scratch: optional
 * [B,2] device floats -- with it the ranges come from one extra launch instead of a whole-plane reduction in every block */
int wmd_mask_level_b(const float* yl, size_t n_yl, const float* yh, float thresh_ratio, int B, int h, int w,
                     const wmd_dilate_spec* specs, int n, float* minmax_scratch, void* stream);

/* ---- Work-list form of a sparse level's mask launch (round 4) --------------------------------------------------------
 * One launch per decoder level, as wmd_mask_level_b / wmd_mask_dilate_multi_b, that ALSO
 *   * compacts, per spec that asks for it, the pixel tiles (tile_h x tile_w) holding at least one set pixel into a work
 *     list for wmd_conv_args.out_tiles: every workgroup finds the active tiles of its own 16x16-cell region (a bit per
 *     tile: wavefront OR-reduction + popcount = the offsets inside its run) and reserves a run of the list with one atomic;
 *     the frame's last workgroup (ticket counter) publishes the frame's total.  List order is unspecified, values never depend
 *     on it.  Tile shapes must nest in the region: tile_h, tile_w divide 16 * up, at most 32 tiles per region.
 *   * publishes the per-frame pixel counts of the specs that ask for it WITHOUT a zero-initialised accumulator of the
 *     caller's and without a copy: accumulators live in `scratch` (all zero at rest: the last workgroup moves them out and
 *     re-arms them) and are written to slot (seq % ring_slots) of the int32 ring `counts` [ring_slots][slot_ints] at
 *     [1 + counts_off + frame * ncounts + (count - 1)]; seq is a launch counter kept in scratch and advanced by the launch
 *     that has `advance` set (the last mask launch of a decoder forward), which also stamps slot[0] = seq + 1 -- so a host
 *     that counts its forwards the same way can fetch the counts of forward k from slot k % ring_slots whenever it wants
 *     them (and verify the stamp), and a forward that nobody asks pays neither a fill, nor a copy, nor a sync.
 *   * takes the LL range from `range_keys` when given: [B][2] order-preserving uint32 keys of (min, max) that the head
 *     kernels' epilogues maintain with atomics (wmd_head_shiftsum_args.range_keys); consumed and re-armed
 *     (0xFFFFFFFF, 0) by this launch.  Else `minmax` [B,2] floats, else every block reduces yl itself.
 * scratch: wmd_mask_level_scratch_ints(B) int32, zero-initialised ONCE by the caller, private to one stream.            */
typedef struct {
    int up, radius;        /* as wmd_dilate_spec                                                                      */
    uint8_t* out;          /* [B, h*up, w*up]                                                                         */
    int count;             /* k > 0: this spec's pixel count is published as count column k - 1 (k <= ncounts <= 8)    */
    int tile_h, tile_w;    /* != 0: build the active-tile list of this spec's mask                                     */
    int32_t* tile_list;    /* [B][ceil(h*up / tile_h) * ceil(w*up / tile_w)]: frame f's tiles fill the head of its segment */
    int32_t* tile_count;   /* [B], overwritten                                                                         */
    const uint8_t* and_mask; /* optional [B, h*up, w*up]: out = dilation AND and_mask.  The never-refilled activation planes
                              of the work-list form hold stale values outside the previous level's support; the reference
                              reads 0 there (sparse_select, KITTI/layers.py:392-400: a pixel absent from the previous index
                              map gathers the prepended zero), so the input mask of upconv(i,0) is lowres AND the previous
                              level's upconv1 mask -- a no-op for thresholded masks, which are nested by construction   */
} wmd_level_spec;

typedef struct {
    int B, h, w;
    const float* yl;       /* [B, n_yl]   threshold form (depth_decoder.py:308-309) ...                                */
    size_t n_yl;
    const float* yh;       /* [B, 3, h, w]                                                                            */
    float thresh_ratio;
    const uint8_t* mask0;  /* ... or an injected base mask [B, h, w] (yl / yh unused)                                  */
    const float* minmax;   /* optional [B, 2]                                                                          */
    uint32_t* range_keys;  /* optional [B, 2], see above                                                               */
    const wmd_level_spec* specs;
    int n;                 /* <= 8                                                                                     */
    int32_t* scratch;
    int32_t* counts;       /* the ring (NULL when ncounts == 0)                                                        */
    int ring_slots, slot_ints, counts_off, ncounts;
    int advance;
} wmd_mask_level_args;
size_t wmd_mask_level_scratch_ints(int B);
int wmd_mask_level_lists(const wmd_mask_level_args* args, void* stream);

typedef struct {
    const uint8_t* mask;   /* [npix]                                                               */
    int npix;
    int32_t* coords;       /* [npix] capacity; receives the active pixel indices in raster order   */
    int32_t* nnz;          /* device scalar                                                        */
} wmd_compact_spec;
/* Stream compaction (mask2idxmap / mask2yx, KITTI/layers.py:371-389) of up to 8 masks in one launch:
 * one workgroup per mask; a lane packs 16 flags into a bit field (popcount), wavefront shuffle prefix sums + an LDS scan
 * over the 16 wavefronts, raster order preserved.                                                   */
int wmd_mask_compact_multi(const wmd_compact_spec* specs, int n, void* stream);
/* batched: specs[i].mask / .coords are [B,npix]; specs[i].nnz points at the count of (frame 0, mask i) of an int32 [B,n]
 * array (the counts of one frame are contiguous)                                                                      */
int wmd_mask_compact_multi_b(const wmd_compact_spec* specs, int n, int B, void* stream);

typedef struct {
    int H, W;              /* output resolution                                                    */
    int C1, up1;           /* source 1: dense [C1tot, H/up1, W/up1]; channels [c1_off, c1_off+C1) used */
    int C1tot, c1_off;
    int C2;                /* source 2 (skip): dense [C2,H,W]; 0 if none                           */
    int Cout, ksize, pad_mode, act;
    float slope;
    const float* x1;
    const float* x2;
    const uint8_t* in_mask;      /* [H,W] support of the virtual input; NULL = everywhere          */
    const int32_t* out_coords;   /* active output pixels                                          */
    const int32_t* out_nnz;
    int max_out;                 /* launch bound: capacity of out_coords                          */
    const float* wp;             /* packed weights [Cout, C1+C2, k, k] (wmd_conv_pack_weights)     */
    const float* bias;
    /* dual-head mode (wp2 != NULL, Cout <= 16): a second filter over channels [c1_off2, c1_off2+C1)
     * of x1; y = out_scale*act(conv) - out_scale*act(conv2)   (depth_decoder.py:276-290)          */
    const float* wp2;
    const float* bias2;
    int c1_off2;
    float out_scale;             /* y = out_scale * act(conv + bias)                              */
    float* y;                    /* dense [Cout,H,W]; written at the active pixels only            */
    int split_waves;             /* 0 = default (2048).  The wavefronts of a block split the input channels of one
                                  * 16-pixel tile while (tiles x slices) stays below this count, and take longer
                                  * channel ranges of separate tiles beyond it (decided on the device from *out_nnz);
                                  * 1 = never split, INT_MAX = always split.  Results agree to fp32 rounding.      */
    int B;                       /* frames decoded by this launch (0 or 1: one).  x1 [B,C1tot,..], x2 [B,C2,H,W], in_mask
                                  * [B,H,W], out_coords [B,max_out], y [B,Cout,H,W]; weights are shared               */
    int nnz_stride;              /* int32 elements between the counts of consecutive frames (out_nnz + b*nnz_stride)  */
} wmd_sparse_conv_args;

/* Gather-GEMM convolution on the active pixels (sparse_conv3x3 / sparse_conv1x1 / sparse_upsample /
 * sparse_select, KITTI/layers.py:337-507) on fp32 MFMA: one wavefront = 16 active pixels x up to 64
 * output channels; the B operand is gathered per lane through the coordinate-padding + mask test.  */
int wmd_sparse_conv(const wmd_sparse_conv_args* args, void* stream);

/* ------------------------------------------------------------------ *
 * Multi-scale loss front-end
 * ------------------------------------------------------------------ */
/* F.interpolate(x, [H,W], mode="bilinear", align_corners) (KITTI/trainer.py:337-338 with align_corners=False;
 * NYUv2/train.py:304-305 with align_corners=True, H = h*2^s) and, when depth != NULL, disp_to_depth
 * (KITTI/layers.py:16-25): depth = 1 / (1/max_depth + (1/min_depth - 1/max_depth) * y).
 * x [N,h,w] -> y [N,H,W] (may be NULL), depth [N,H,W] (may be NULL).                                        */
int wmd_upsample_bilinear_fwd(const float* x, float* y, float* depth, int N, int h, int w, int H, int W,
                              int align_corners, float min_depth, float max_depth, void* stream);
/* Adjoint: dx [N,h,w] = B^T (dy + ddepth * d depth/d y); `depth` is the forward output (needed with ddepth). */
int wmd_upsample_bilinear_bwd(const float* dy, const float* ddepth, const float* depth, float* dx, int N, int h,
                              int w, int H, int W, int align_corners, float min_depth, float max_depth, void* stream);

/* ------------------------------------------------------------------ *
 * Data-parallel gradient exchange (new: the reference is single-GPU, trainer.py:45)
 * ------------------------------------------------------------------ */
typedef struct wmd_comm wmd_comm;
/* unique_id: 128 bytes produced by rank 0 with wmd_comm_unique_id and broadcast by the caller */
int wmd_comm_unique_id(void* unique_id_128);
int wmd_comm_init(wmd_comm** comm, const void* unique_id_128, int world, int rank);
/* in-place sum all-reduce of `n` floats then scale by `scale` (1/world), on `stream` */
int wmd_comm_allreduce(wmd_comm* comm, float* buf, size_t n, float scale, void* stream);
/* in-place broadcast of n floats from rank `root` (initial parameter / BatchNorm-buffer synchronisation of the replicas) */
int wmd_comm_broadcast(wmd_comm* comm, float* buf, size_t n, int root, void* stream);
/* what the communicator itself reports: the RCCL version code (ncclGetVersion), ncclCommCount, ncclCommUserRank */
int wmd_comm_info(wmd_comm* comm, int* rccl_version, int* world, int* rank);
int wmd_comm_destroy(wmd_comm* comm);

/* ------------------------------------------------------------------ *
 * Evaluation arithmetic behind the reference's accuracy claims; batches stay in HBM.
 * ------------------------------------------------------------------ */

/* Per-image chain of KITTI/evaluate_depth.py:268-307 for B images of one size:
 *   pred_disp [B,h,w] --cv2.resize (bilinear, half-pixel centres, edge clamp)--> [B,H,W] --> depth = pred_scale / disp
 *   mask_mode 1: Eigen split: min_depth < gt < max_depth inside the Garg crop (:284-290); 0: gt > 0 (:293)
 *   median_scaling != 0: depth *= median(gt[mask]) / median(depth[mask])  (:298-301; np.median = mean of the two middle values)
 *   depth = clamp(depth, min_depth, max_depth) (:304-305); compute_errors (:50-68)
 * out [B,9] = abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, log10, n_valid  (a row of NaN + 0 when the mask is empty);
 * the applied ratio of image b is left in workspace (see wmd_eval_workspace_floats) for the caller to read.        */
typedef struct {
    int B, h, w, H, W;
    float min_depth, max_depth;   /* 1e-3, 80 (evaluate_depth.py:85-86)                    */
    int mask_mode;
    float pred_scale;             /* pred_depth_scale_factor (5.4 for stereo, :35)         */
    int median_scaling;
    const float* pred_disp;
    const float* gt_depth;
    float* out;
    float* workspace;             /* wmd_eval_workspace_floats(B, H, W)                    */
    size_t workspace_floats;
} wmd_eval_kitti_args;
size_t wmd_eval_workspace_floats(int B, int H, int W);
int wmd_eval_kitti(const wmd_eval_kitti_args* args, void* stream);

/* compute_errors (KITTI/evaluate_depth.py:50-68) and compute_errors_nyu (NYUv2/utils.py:85-98) on B prepared
 * arrays of n_per_image positive values each: out9 [B,9] as above (NYUv2 reads abs_rel, rmse, log10, a1..a3).
 * Entries with gt < 0 are skipped (masked).                                                                      */
int wmd_eval_errors(const float* pred, const float* gt, int B, size_t n_per_image, float* out9, void* stream);

/* batch_post_process_disparity (KITTI/evaluate_depth.py:71-79) with the [:, :, ::-1] flip of the second operand
 * (:204) fused: r_disp is the raw prediction for the mirrored image.  l_disp, r_disp, out: [B,h,w].              */
int wmd_flip_postprocess(const float* l_disp, const float* r_disp, float* out, int B, int h, int w, void* stream);

/* ------------------------------------------------------------------ *
 * Photometric loss stack of the KITTI trainer, forward and backward
 * ------------------------------------------------------------------ */

/* SSIM (KITTI/layers.py:281-311) and compute_reprojection_loss (KITTI/trainer.py:393-405).  x = pred, y = target, [B,C,H,W].
 * mode 0: out [B,C,H,W] = clamp((1 - SSIM(x, y)) / 2, 0, 1)                      (the SSIM module)
 * mode 1: out [B,1,H,W] = w_ssim * mean_c(mode 0) + w_l1 * mean_c |y - x|        (0.85 / 0.15 in the trainer)
 * Backward: g has the shape of out; dx and/or dy [B,C,H,W] (either may be NULL); workspace = 3*B*C*H*W floats.       */
int wmd_ssim_fwd(const float* x, const float* y, float* out, int B, int C, int H, int W, int mode, float w_ssim, float w_l1,
                 void* stream);
size_t wmd_ssim_bwd_workspace_floats(int B, int C, int H, int W);
int wmd_ssim_bwd(const float* x, const float* y, const float* g, float* dx, float* dy, int B, int C, int H, int W, int mode,
                 float w_ssim, float w_l1, float* workspace, size_t workspace_floats, void* stream);

/* BackprojectDepth -> Project3D -> F.grid_sample(src, grid, padding_mode="border") (KITTI/layers.py:176-229,
 * KITTI/trainer.py:352-372) in one kernel: out[b,c,y,x] = bilinear(src[b,c], project(K T, depth[b,y,x] * inv_K (x,y,1))).
 * K, inv_K, T: [B,4,4] row-major.  Backward returns d(depth) [B,1,H,W] and dT [B,4,4]; the source frame and the
 * intrinsics are constants of the loss (no gradient), as in the trainer.                                              */
typedef struct {
    int B, C, H, W;      /* target grid = depth map size; C colour channels                                          */
    int Hs, Ws;          /* source frame size (equal to H, W in the trainer)                                         */
    float eps;           /* Project3D eps, 1e-7                                                                       */
    const float* src;    /* [B,C,Hs,Ws] */
    const float* depth;  /* [B,1,H,W]   */
    const float* K;
    const float* inv_K;
    const float* T;
} wmd_warp_args;
int wmd_warp_fwd(const wmd_warp_args* args, float* out, void* stream);
size_t wmd_warp_bwd_workspace_floats(const wmd_warp_args* args);
int wmd_warp_bwd(const wmd_warp_args* args, const float* grad_out, float* ddepth, float* dT, float* workspace,
                 size_t workspace_floats, void* stream);

/* get_smooth_loss (KITTI/layers.py:238-252): out[0] = mean |dx disp| exp(-gamma mean_c |dx img|) + the same along y.
 * disp [B,1,H,W], img [B,C,H,W]; backward w.r.t. disp only (grad_out: 1 float on the device).                          */
size_t wmd_smooth_workspace_floats(int B, int H, int W);
int wmd_smooth_fwd(const float* disp, const float* img, float* out, int B, int C, int H, int W, float gamma, float* workspace,
                   size_t workspace_floats, void* stream);
int wmd_smooth_bwd(const float* disp, const float* img, const float* grad_out, float* ddisp, int B, int C, int H, int W,
                   float gamma, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WMD_H */
