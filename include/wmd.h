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
 *   [{"kernel": name, "calls": n, "ms": total, "flops": algorithmic FLOPs, "bytes": algorithmic bytes}, ...]
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
    int tune_ksplit;    /* 0: cost model; k>0: force k-way split of the Cin reduction    */
} wmd_conv_args;

/* Fused  upsample(x1) ++ x2  ->  pad  ->  conv kxk  ->  + bias  ->  activation.
 * Replaces ConvBlock/Conv3x3/Conv1x1 (KITTI/layers.py:120-173), `upsample` (:233-236),
 * the skip `torch.cat` (depth_decoder.py:146-149), NYUv2 Conv3x3/UpSampleBlock
 * (NYUv2/networks/layers.py:11-32,57-67).                                              */
int wmd_conv_fwd(const wmd_conv_args* args, void* stream);

/* Suggested workspace size (floats) for wmd_conv_fwd on this problem. */
size_t wmd_conv_fwd_workspace_floats(const wmd_conv_args* args);

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
} wmd_conv_wgrad_args;

size_t wmd_conv_wgrad_workspace_floats(const wmd_conv_wgrad_args* args);
/* Weight + bias gradient (deterministic two-stage split over the B*H*W reduction). */
int wmd_conv_wgrad(const wmd_conv_wgrad_args* args, void* stream);

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
} wmd_head_args;

/* Replaces Conv3x3(C,3|1) + Sigmoid + the 2^(s-1)(sigma+ - sigma-) combine of
 * DepthWaveProgressiveDecoder.get_coefficients (depth_decoder.py:126-136) and the
 * NYUv2 wave1_ll/wave{1,2,3} convolutions (densedepth_decoder.py:106-115).             */
int wmd_head3x3_fwd(const wmd_head_args* args, void* stream);

/* ------------------------------------------------------------------ *
 * Sparse (threshold-gated) decoder path, batch 1
 * ------------------------------------------------------------------ */

/* min and max of x[0..n) -> out2[0]=min, out2[1]=max  (depth_decoder.py:308 yl.max()-yl.min()) */
int wmd_minmax(const float* x, size_t n, float* out2, void* workspace, size_t workspace_floats, void* stream);

/* mask[y,x] = max_b |yh[b,y,x]| > (minmax[1]-minmax[0])*thresh_ratio   (depth_decoder.py:308-309)
 * yh [3,h,w]; mask uint8 [h,w].  thresh_ratio < 0 or force_all != 0 -> all ones.        */
int wmd_mask_threshold(const float* yh, const float* minmax, float thresh_ratio, int force_all,
                       uint8_t* mask, int h, int w, void* stream);

/* out[y,x] = max over the (2r+1)^2 window of in[(y,x)/up]  (MaxPool2d(2r+1,1,r) of an optionally
 * nearest-upsampled mask, depth_decoder.py:311-319).  in [h,w]; out [h*up, w*up].      */
int wmd_mask_dilate(const uint8_t* in, uint8_t* out, int h, int w, int up, int radius, void* stream);

/* Raster-order stream compaction (mask2idxmap, KITTI/layers.py:382-389): idx[p] = rank of p among
 * active pixels or -1; *nnz_dev = count.  Wavefront ballot + popcount prefix sums.
 * workspace: >= wmd_mask_compact_workspace_bytes(h*w) bytes.                            */
size_t wmd_mask_compact_workspace_bytes(int npix);
int wmd_mask_compact(const uint8_t* mask, int32_t* idxmap, int32_t* coords /* [nnz] packed y*w+x, may be NULL */,
                     int32_t* nnz_dev, int npix, void* workspace, size_t workspace_bytes, void* stream);

typedef struct {
    int H, W;           /* resolution of the output mask                                */
    int C1, up1;        /* compact source 1: vals1 [nnz1, C1] pixel-major, idxmap1 [H/up1, W/up1] */
    int C2;             /* dense skip source [C2,H,W] (may be 0)                        */
    int Cout, ksize, pad_mode, act; float slope;
    int Cmid;           /* >0: a fused leading 1x1 (C1->Cmid, LeakyReLU(slope_mid)) before the kxk, as
                           sparse_conv3x3 does for nn.Sequential heads (layers.py:426-431) */
    float slope_mid;
    const float* vals1; const int32_t* idxmap1;
    const float* x2;
    const int32_t* coords_out; const int32_t* nnz_out;   /* active output pixels         */
    int max_nnz_out;    /* launch bound (capacity of coords_out / vals_out)             */
    const float* w; const float* bias;                   /* [Cout, C1+C2 (or Cmid), k, k] */
    const float* w_mid; const float* bias_mid;           /* [Cmid, C1]                   */
    float* vals_out;    /* compact [nnz_out, Cout]                                      */
    float* dense_out;   /* optional dense scatter target [Cout,H,W] (pre-zeroed)        */
    float dense_scale;
} wmd_sparse_conv_args;

/* Gather-GEMM convolution on active pixels (sparse_conv3x3 / sparse_conv1x1 / sparse_upsample /
 * sparse_select, KITTI/layers.py:337-507): neighbours outside the input mask read zero, the
 * index map is padded reflect/constant/replicate on indices (layers.py:444).            */
int wmd_sparse_conv(const wmd_sparse_conv_args* args, void* stream);

/* ------------------------------------------------------------------ *
 * Data-parallel gradient exchange (new: the reference is single-GPU, trainer.py:45)
 * ------------------------------------------------------------------ */
typedef struct wmd_comm wmd_comm;
/* unique_id: 128 bytes produced by rank 0 with wmd_comm_unique_id and broadcast by the caller */
int wmd_comm_unique_id(void* unique_id_128);
int wmd_comm_init(wmd_comm** comm, const void* unique_id_128, int world, int rank);
/* in-place sum all-reduce of `n` floats then scale by `scale` (1/world), on `stream` */
int wmd_comm_allreduce(wmd_comm* comm, float* buf, size_t n, float scale, void* stream);
int wmd_comm_destroy(wmd_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* WMD_H */
