// Backward of the fused decoder convolution (autograd of ConvBlock/Conv3x3/Conv1x1 + upsample + cat + pad;
// the reference gets it implicitly from torch.autograd, KITTI/trainer.py:211, NYUv2/train.py:327).
//
//   forward:  z = W * P(x1, x2),   P = pad o concat o nearest-upsample   (a linear gather)
//   dgrad:    dP = W^T (*) dz on the padded (H+2)x(W+2) domain  -> conv_fwd_kernel fed with dz, the
//             transposed+flipped weight image and shift1 = 1;   dx = P^T dP  -> conv_dgrad_fold_kernel
//             (reflect/replicate border accumulation, channel split, 2x2 sum for the upsampled source)
//   wgrad:    dW[co,ci,t] = sum_{b,y,x} dz[b,co,y,x] * P[b,ci,y+ky,x+kx]  -> conv_wgrad_kernel: an MFMA GEMM
//             with M = co, N = (ci,tap), K = pixels; every block owns a (co-tile, ci-tile) pair and a slice
//             of the pixel tiles, accumulates in registers and writes ONE partial; a second kernel sums the
//             partials (deterministic two-stage reduction, no atomics).
//   dbias:    row sums of the dz tiles the wgrad blocks already hold in LDS, reduced with the weight partials.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "wmd_conv_common.h"

namespace wmd {


int run_conv(const wmd_conv_args* g, int shift1, int H1, int W1, void* stream);

// ------------------------------------------------------------------------------------------------
// dgrad stage 2: adjoint of pad + concat + upsample
// ------------------------------------------------------------------------------------------------
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte global load

// One (image, channel) plane: out[y, x] = sum of the u x u padded-domain gradients that the forward gather read from
// source pixel (y, x), plus -- on the (rare, divergent) border lines -- the halo rows / columns that reflect or
// replicate padding mirrored onto it.  VEC: a thread owns 4 consecutive padded-domain columns (4/U outputs).
template <int U, bool VEC>
__device__ __forceinline__ void fold_plane(const float* __restrict__ gp, float* __restrict__ op, int H, int W, int pad_mode,
                                           int halo, const float* __restrict__ gate, int gate_act, float gate_slope) {
    constexpr int NX = VEC ? 4 : U;       // padded-domain columns per thread
    constexpr int VX = NX / U;            // outputs per thread
    const int Wp = W + 2 * halo;
    const int h_ = H / U, w_ = W / U, wq = w_ / VX;
    const int lo_src = pad_mode == WMD_PAD_REFLECT ? 1 : 0, hi_src = W - (pad_mode == WMD_PAD_REFLECT ? 2 : 1);
    const int lo_row = lo_src, hi_row = H - (pad_mode == WMD_PAD_REFLECT ? 2 : 1);
    const bool folds = halo && pad_mode != WMD_PAD_ZERO;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < h_ * wq; i += gridDim.x * blockDim.x) {
        const int y = i / wq, xq = i - y * wq;
        const int X0 = xq * NX;
        float o[VX];
#pragma unroll
        for (int e = 0; e < VX; ++e) o[e] = 0.f;
        auto add_row = [&](int r) {
            const float* p = gp + (size_t)r * Wp;
            float v[NX];
            if constexpr (VEC) {
                const f32x4u q = *reinterpret_cast<const f32x4u*>(p + X0 + halo);
                v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
            } else {
#pragma unroll
                for (int e = 0; e < NX; ++e) v[e] = p[X0 + halo + e];
            }
#pragma unroll
            for (int e = 0; e < NX; ++e) {
                float s = v[e];
                if (folds && X0 + e == lo_src) s += p[0];
                if (folds && X0 + e == hi_src) s += p[W + 1];
                o[e / U] += s;
            }
        };
#pragma unroll
        for (int dy = 0; dy < U; ++dy) {
            const int Y = y * U + dy;
            add_row(Y + halo);
            if (folds && Y == lo_row) add_row(0);
            if (folds && Y == hi_row) add_row(H + 1);
        }
        float* dst = op + (size_t)y * w_ + xq * VX;
        if (gate) {   // the source plane is itself an activation output: hand its producer dz = dx * f'(x)
            const float* gq = gate + (size_t)y * w_ + xq * VX;
            float gv[VX];
            if constexpr (VX == 4) {
                const float4 q4 = *reinterpret_cast<const float4*>(gq);
                gv[0] = q4.x, gv[1] = q4.y, gv[2] = q4.z, gv[3] = q4.w;
            } else if constexpr (VX == 2) {
                const float2 q2 = *reinterpret_cast<const float2*>(gq);
                gv[0] = q2.x, gv[1] = q2.y;
            } else {
                gv[0] = gq[0];
            }
#pragma unroll
            for (int e = 0; e < VX; ++e) o[e] *= act_deriv(gv[e], gate_act, gate_slope);
        }
        if constexpr (VX == 4) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else if constexpr (VX == 2) *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
        else dst[0] = o[0];
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void conv_dgrad_fold_kernel(const float* __restrict__ g, float* __restrict__ dx1,
                                                              float* __restrict__ dx2, int B, int C1, int C2, int H, int W,
                                                              int up1, int pad_mode, int halo, const float* __restrict__ x1_fwd,
                                                              int x1_act, float x1_slope) {
    // blockIdx.y = (image, channel) plane of the padded-domain gradient
    const int Cin = C1 + C2;
    const int b = blockIdx.y / Cin, ch = blockIdx.y % Cin;
    const bool first = ch < C1;
    float* out = first ? dx1 : dx2;
    if (!out) return;
    const int u = first ? up1 : 1;
    const float* gp = g + (size_t)blockIdx.y * (H + 2 * halo) * (W + 2 * halo);
    const size_t ooff = ((size_t)b * (first ? C1 : C2) + (first ? ch : ch - C1)) * (H / u) * (W / u);
    float* op = out + ooff;
    const float* gate = (first && x1_fwd && x1_act != WMD_ACT_NONE) ? x1_fwd + ooff : nullptr;
    if (u == 2) fold_plane<2, VEC>(gp, op, H, W, pad_mode, halo, gate, x1_act, x1_slope);
    else fold_plane<1, VEC>(gp, op, H, W, pad_mode, halo, gate, x1_act, x1_slope);
}

// ------------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wg_dma4(__amdgpu_buffer_rsrc_t r, lds_ptr_t dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 4, voff, soff, 0, 0);
}

// Block = WM x WN waves. Wave (wm, wn) owns MR out-channel tiles (16 each) x NC 16-input-channel groups x TAPS.
// Pixel tile = TH x TW (TW % 4 == 0): the MFMA K index walks 4 consecutive pixels of a row.
// Both operand tiles are gathered by LDS-DMA (position-linear rows), double buffered across pixel tiles.
template <int TH, int TW, int MR, int NC, int WM, int WN, int TAPS>
struct WgradTile {
    static constexpr int NT = WM * WN * 64;
    static constexpr int HALO = TAPS == 9 ? 1 : 0;
    static constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    static constexpr int NPIX = TH * TW;
    static constexpr int NPATCH = PH * PW;
    static constexpr int COT = WM * MR * 16;        // out channels per block
    static constexpr int CIT = WN * NC * 16;        // in channels per block
    // row strides: >= the DMA span (whole 64-lane pieces) and == 2 (mod 32) so that lanes (i = l&15, k = l>>4) of a
    // 32-lane ds_read_b32 group hit banks 2i + k, all distinct
    static constexpr int SPAN_A = ((NPIX + 63) / 64) * 64;
    static constexpr int SPAN_B = ((NPATCH + 63) / 64) * 64;
    static constexpr int SA = ((SPAN_A - 2 + 31) / 32) * 32 + 2;
    static constexpr int SB = ((SPAN_B - 2 + 31) / 32) * 32 + 2;
    static constexpr int BUF = COT * SA + CIT * SB;
    static constexpr int LDS_FLOATS = 2 * BUF;
    static constexpr int PA = SPAN_A / 64, PB = SPAN_B / 64;   // 64-lane DMA pieces per row
    static_assert(TW % 4 == 0, "pixel quads must not straddle rows");
};

template <int TH, int TW, int MR, int NC, int WM, int WN, int TAPS>
__global__ __launch_bounds__(WM* WN * 64) void conv_wgrad_kernel(const WgradKArgs a) {
    using T = WgradTile<TH, TW, MR, NC, WM, WN, TAPS>;
    constexpr int HALO = T::HALO, PW = T::PW, SA = T::SA, SB = T::SB, NPIX = T::NPIX;
    constexpr int NWAVES = WM * WN;
    constexpr unsigned kOOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int co0 = blockIdx.y * T::COT, ci0 = blockIdx.x * T::CIT;
    const int split = blockIdx.z;
    const int H = a.H, W = a.W;
    const size_t plane = (size_t)H * W, plane1 = (size_t)a.H1 * a.W1;
    const unsigned pbz = (unsigned)(plane * 4), pb1 = (unsigned)(plane1 * 4);

    f32x4 acc[MR][NC][TAPS];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[m][n][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;  // bias partial: thread c < COT sums row c of the dz tiles (ci-tile 0 blocks only)

    const int per = (a.ntiles + a.nsplit - 1) / a.nsplit;
    const int t_begin = split * per, t_end = min(t_begin + per, a.ntiles);

    // DMA of one pixel tile into buffer `buf`: rows of dz (out channels) and of the gathered input patch.
    // Rows are handed round-robin to the waves; a row is PA (PB) pieces of 64 consecutive positions.
    auto stage = [&](int tile, int buf) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int b = t / a.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        float* bufA = lds + buf * T::BUF;
        float* bufB = bufA + T::COT * SA;
        const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.dz + (size_t)b * a.Cout * plane), 0, (int)(a.Cout * plane * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x1 + (size_t)b * a.C1 * plane1), 0, (int)(a.C1 * plane1 * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x2 ? a.x2 + (size_t)b * a.C2 * plane : a.x1), 0, (int)(a.C2 * plane * 4), 0x00020000);
        // per-lane byte offsets inside a channel plane for the PA / PB pieces of a row (same for every row)
        unsigned oz[T::PA], o1[T::PB], o2[T::PB];
#pragma unroll
        for (int i = 0; i < T::PA; ++i) {
            const int p = i * 64 + lane;
            const int oy = y0 + p / TW, ox = x0 + p % TW;
            oz[i] = (p < NPIX && oy < H && ox < W) ? (unsigned)(oy * W + ox) * 4u : kOOB;
        }
#pragma unroll
        for (int i = 0; i < T::PB; ++i) {
            const int p = i * 64 + lane;
            int gy = y0 + p / PW - HALO, gx = x0 + p % PW - HALO;
            bool ok = p < T::NPATCH;
            if (HALO) {
                ok = pad_coord(gy, H, a.pad_mode) && ok;
                ok = pad_coord(gx, W, a.pad_mode) && ok;
            }
            ok = ok && gy >= 0 && gx >= 0 && gy < H && gx < W;
            gy = min(max(gy, 0), H - 1);
            gx = min(max(gx, 0), W - 1);
            o2[i] = ok ? (unsigned)(gy * W + gx) * 4u : kOOB;
            o1[i] = ok ? (unsigned)((gy / a.up1) * a.W1 + gx / a.up1) * 4u : kOOB;
        }
        for (int c = wave; c < T::COT; c += NWAVES) {       // wave-uniform rows
            const int co = co0 + c;
            const unsigned so = (unsigned)min(co, a.Cout - 1) * pbz;
#pragma unroll
            for (int i = 0; i < T::PA; ++i) wg_dma4(rz, (lds_ptr_t)(bufA + c * SA + i * 64), co < a.Cout ? oz[i] : kOOB, so);
        }
        for (int c = wave; c < T::CIT; c += NWAVES) {
            const int ci = ci0 + c;
            const bool from1 = ci < a.C1;
            const unsigned so = from1 ? (unsigned)ci * pb1 : (unsigned)min(max(ci - a.C1, 0), max(a.C2 - 1, 0)) * pbz;
#pragma unroll
            for (int i = 0; i < T::PB; ++i) {
                const unsigned vo = ci < a.Cin ? (from1 ? o1[i] : o2[i]) : kOOB;
                if (from1) wg_dma4(r1, (lds_ptr_t)(bufB + c * SB + i * 64), vo, so);
                else wg_dma4(r2, (lds_ptr_t)(bufB + c * SB + i * 64), vo, so);
            }
        }
    };

    if (t_begin < t_end) stage(t_begin, 0);
    __syncthreads();

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int buf = (tile - t_begin) & 1;
        if (tile + 1 < t_end) stage(tile + 1, buf ^ 1);
        const float* ldsA = lds + buf * T::BUF;
        const float* ldsB = ldsA + T::COT * SA;
        const float* pa = ldsA + (wm * MR * 16 + (lane & 15)) * SA + (lane >> 4);
        const float* pb = ldsB + (wn * NC * 16 + (lane & 15)) * SB + (lane >> 4);
        // K-steps of 4 pixels, fully unrolled (every LDS offset is an immediate) and software-pipelined one step deep: the
        // fragments of step q+1 are requested before the MFMAs of step q are issued and sched_barrier pins that order --
        // with 2 waves per SIMD an LDS round trip in front of every group of MFMAs was the main loss (59-77 TFLOP/s).
        constexpr int QS = NPIX / 4;
        float af[2][MR], bf[2][NC][TAPS];
        auto fetch = [&](int qs) {
            const int q = qs * 4, py = q / TW, px = q % TW, sl = qs & 1;
#pragma unroll
            for (int m = 0; m < MR; ++m) af[sl][m] = pa[m * 16 * SA + q];
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int ky = TAPS == 9 ? t / 3 : 0, kx = TAPS == 9 ? t % 3 : 0;
                    bf[sl][n][t] = pb[n * 16 * SB + (py + ky) * PW + px + kx];
                }
        };
        fetch(0);
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
            if (qs + 1 < QS) fetch(qs + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int m = 0; m < MR; ++m)
                        acc[m][n][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[qs & 1][m], bf[qs & 1][n][t], acc[m][n][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (a.want_bias && blockIdx.x == 0 && tid < T::COT) {
            const float* row = ldsA + tid * SA;
            float s = 0.f;
            for (int p = 0; p < NPIX; ++p) s += row[p];
            bsum += s;
        }
        __syncthreads();
    }

    // ---- write this block's partial: D row = out channel (lane>>4)*4+r, D col = input channel lane&15
    const size_t nw = (size_t)a.Cout * a.Cin * TAPS;
    float* out = a.partial + (size_t)split * (nw + a.Cout);
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        const int ci = ci0 + (wn * NC + n) * 16 + (lane & 15);
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wm * MR + m) * 16 + (lane >> 4) * 4 + r;
                if (co < a.Cout && ci < a.Cin) {
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) out[((size_t)co * a.Cin + ci) * TAPS + t] = acc[m][n][t][r];
                }
            }
    }
    if (a.want_bias && blockIdx.x == 0 && tid < T::COT && co0 + tid < a.Cout) out[nw + co0 + tid] = bsum;
}

// ------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) weight gradient.
//   forward (conv_wino_kernel):  Y = A^T [ sum_ci U (.) V ] A,  U = G g G^T,  V = B^T d B   per 2x2 output tile
//   =>  dU_xi[co,ci] = sum_tiles dM_xi[tile,co] * V_xi[tile,ci],  dM = A dY A^T (2x2 -> 4x4),   dg = G^T dU G
// In the transformed domain each of the 16 positions xi is an independent GEMM with K = tiles: 16 MFMAs per 4 tiles
// (= 16 pixels) and (co-tile, ci-tile) pair instead of the 36 of the direct form (4 K-steps x 9 taps).  A lane owns one
// channel (l & 15) and one tile of the K-step (l >> 4) for BOTH operands: it reads the tile's 2x2 dz values and the 4x4
// input patch from LDS at immediate offsets and transforms them in registers (12 + 32 adds).  Staging (LDS-DMA gather of
// dz rows and of the padded / upsampled / concatenated patch rows, double buffered across pixel tiles), the split over
// pixel tiles and the bias row sums are those of conv_wgrad_kernel; partials are [split][16][Cout*Cin] (+ [Cout] bias) and
// wgrad_wino_reduce_kernel sums them and applies G^T . G.
// ------------------------------------------------------------------------------------------------
template <int TH, int TW, int MR, int NC, int WM, int WN>
struct WgradWinoTile {
    static constexpr int NT = WM * WN * 64;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int NPIX = TH * TW, NPATCH = PH * PW;
    static constexpr int COT = WM * MR * 16, CIT = WN * NC * 16;
    static constexpr int TXW = TW / 2, NT2 = (TH / 2) * TXW, KS = NT2 / 4;
    // row strides == 2 (mod 32), no rounding to whole DMA pieces: the last piece of a row is exec-masked
    static constexpr int SA = ((NPIX - 2 + 31) / 32) * 32 + 2;
    static constexpr int SB = ((NPATCH - 2 + 31) / 32) * 32 + 2;
    static constexpr int PA = (NPIX + 63) / 64, PB = (NPATCH + 63) / 64;
    static constexpr int BUF = COT * SA + CIT * SB;
    static constexpr int LDS_FLOATS = 2 * BUF;
    static_assert(TH % 2 == 0 && TW % 8 == 0, "whole 2x2 tiles; the four tiles of a K-step stay in one tile row");
    static_assert(NT2 % 4 == 0, "whole K-steps");
};

template <int TH, int TW, int MR, int NC, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_wgrad_wino_kernel(const WgradKArgs a) {
    using T = WgradWinoTile<TH, TW, MR, NC, WM, WN>;
    constexpr int PW = T::PW, SA = T::SA, SB = T::SB, NPIX = T::NPIX;
    constexpr int NWAVES = WM * WN;
    constexpr unsigned kOOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int co0 = blockIdx.y * T::COT, ci0 = blockIdx.x * T::CIT;
    const int split = blockIdx.z;
    const int H = a.H, W = a.W;
    const size_t plane = (size_t)H * W, plane1 = (size_t)a.H1 * a.W1;
    const unsigned pbz = (unsigned)(plane * 4), pb1 = (unsigned)(plane1 * 4);

    f32x4 acc[16][MR][NC];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NC; ++n) acc[xi][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    const int per = (a.ntiles + a.nsplit - 1) / a.nsplit;
    const int t_begin = split * per, t_end = min(t_begin + per, a.ntiles);

    auto stage = [&](int tile, int buf) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int b = t / a.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        float* bufA = lds + buf * T::BUF;
        float* bufB = bufA + T::COT * SA;
        const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.dz + (size_t)b * a.Cout * plane), 0, (int)(a.Cout * plane * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x1 + (size_t)b * a.C1 * plane1), 0, (int)(a.C1 * plane1 * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x2 ? a.x2 + (size_t)b * a.C2 * plane : a.x1), 0, (int)(a.C2 * plane * 4), 0x00020000);
        unsigned oz[T::PA], o1[T::PB], o2[T::PB];
#pragma unroll
        for (int i = 0; i < T::PA; ++i) {
            const int p = i * 64 + lane;
            const int oy = y0 + p / TW, ox = x0 + p % TW;
            oz[i] = (p < NPIX && oy < H && ox < W) ? (unsigned)(oy * W + ox) * 4u : kOOB;
        }
#pragma unroll
        for (int i = 0; i < T::PB; ++i) {
            const int p = i * 64 + lane;
            int gy = y0 + p / PW - 1, gx = x0 + p % PW - 1;
            bool ok = p < T::NPATCH;
            ok = pad_coord(gy, H, a.pad_mode) && ok;
            ok = pad_coord(gx, W, a.pad_mode) && ok;
            ok = ok && gy >= 0 && gx >= 0 && gy < H && gx < W;
            gy = min(max(gy, 0), H - 1);
            gx = min(max(gx, 0), W - 1);
            o2[i] = ok ? (unsigned)(gy * W + gx) * 4u : kOOB;
            o1[i] = ok ? (unsigned)((gy / a.up1) * a.W1 + gx / a.up1) * 4u : kOOB;
        }
        for (int c = wave; c < T::COT; c += NWAVES) {       // wave-uniform rows
            const int co = co0 + c;
            const unsigned so = (unsigned)min(co, a.Cout - 1) * pbz;
#pragma unroll
            for (int i = 0; i < T::PA; ++i)
                if (T::PA * 64 == NPIX || i * 64 + lane < NPIX)   // partial last piece: exec-masked
                    wg_dma4(rz, (lds_ptr_t)(bufA + c * SA + i * 64), co < a.Cout ? oz[i] : kOOB, so);
        }
        for (int c = wave; c < T::CIT; c += NWAVES) {
            const int ci = ci0 + c;
            const bool from1 = ci < a.C1;
            const unsigned so = from1 ? (unsigned)ci * pb1 : (unsigned)min(max(ci - a.C1, 0), max(a.C2 - 1, 0)) * pbz;
#pragma unroll
            for (int i = 0; i < T::PB; ++i) {
                const unsigned vo = ci < a.Cin ? (from1 ? o1[i] : o2[i]) : kOOB;
                if (T::PB * 64 == T::NPATCH || i * 64 + lane < T::NPATCH) {
                    if (from1) wg_dma4(r1, (lds_ptr_t)(bufB + c * SB + i * 64), vo, so);
                    else wg_dma4(r2, (lds_ptr_t)(bufB + c * SB + i * 64), vo, so);
                }
            }
        }
    };

    if (t_begin < t_end) stage(t_begin, 0);
    __syncthreads();

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int buf = (tile - t_begin) & 1;
        if (tile + 1 < t_end) stage(tile + 1, buf ^ 1);
        const float* ldsA = lds + buf * T::BUF;
        const float* ldsB = ldsA + T::COT * SA;
        // lane = (channel l & 15, tile l >> 4 of the K-step): the tile's column offset 2 * (l >> 4) lives in the base pointers
        const float* pa = ldsA + (wm * MR * 16 + (lane & 15)) * SA + 2 * (lane >> 4);
        const float* pb = ldsB + (wn * NC * 16 + (lane & 15)) * SB + 2 * (lane >> 4);
        float dzr[2][MR][4], xr[2][NC][16];
        auto fetch = [&](int ks) {
            const int t0 = ks * 4, trow = t0 / T::TXW, tcol = t0 % T::TXW, sl = ks & 1;
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) dzr[sl][m][r * 2 + c] = pa[m * 16 * SA + (2 * trow + r) * TW + 2 * tcol + c];
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) xr[sl][n][r * 4 + c] = pb[n * 16 * SB + (2 * trow + r) * PW + 2 * tcol + c];
        };
        fetch(0);
#pragma unroll
        for (int ks = 0; ks < T::KS; ++ks) {
            const int sl = ks & 1;
            if (ks + 1 < T::KS) fetch(ks + 1);
            // dM = A dY A^T,  A = [[1,0],[1,1],[1,-1],[0,-1]]
            float dm[MR][16], v[NC][16];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const float d00 = dzr[sl][m][0], d01 = dzr[sl][m][1], d10 = dzr[sl][m][2], d11 = dzr[sl][m][3];
                const float t[4][2] = {{d00, d01}, {d00 + d10, d01 + d11}, {d00 - d10, d01 - d11}, {-d10, -d11}};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dm[m][r * 4 + 0] = t[r][0];
                    dm[m][r * 4 + 1] = t[r][0] + t[r][1];
                    dm[m][r * 4 + 2] = t[r][0] - t[r][1];
                    dm[m][r * 4 + 3] = -t[r][1];
                }
            }
            // V = B^T d B (as in conv_wino_kernel)
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                float tr[16];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    tr[0 * 4 + cc] = xr[sl][n][0 * 4 + cc] - xr[sl][n][2 * 4 + cc];
                    tr[1 * 4 + cc] = xr[sl][n][1 * 4 + cc] + xr[sl][n][2 * 4 + cc];
                    tr[2 * 4 + cc] = xr[sl][n][2 * 4 + cc] - xr[sl][n][1 * 4 + cc];
                    tr[3 * 4 + cc] = xr[sl][n][1 * 4 + cc] - xr[sl][n][3 * 4 + cc];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[n][r * 4 + 0] = tr[r * 4 + 0] - tr[r * 4 + 2];
                    v[n][r * 4 + 1] = tr[r * 4 + 1] + tr[r * 4 + 2];
                    v[n][r * 4 + 2] = tr[r * 4 + 2] - tr[r * 4 + 1];
                    v[n][r * 4 + 3] = tr[r * 4 + 1] - tr[r * 4 + 3];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
#pragma unroll
                for (int n = 0; n < NC; ++n)
#pragma unroll
                    for (int m = 0; m < MR; ++m)
                        acc[xi][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(dm[m][xi], v[n][xi], acc[xi][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (a.want_bias && blockIdx.x == 0 && tid < T::COT) {
            const float* row = ldsA + tid * SA;
            float s = 0.f;
            for (int p = 0; p < NPIX; ++p) s += row[p];
            bsum += s;
        }
        __syncthreads();
    }

    // partial [split][16][Cout*Cin] (+ [Cout] bias sums): D row = out channel (lane>>4)*4+r, D col = input channel lane&15
    const size_t nwc = (size_t)a.Cout * a.Cin;
    float* out = a.partial + (size_t)split * (16 * nwc + a.Cout);
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        const int ci = ci0 + (wn * NC + n) * 16 + (lane & 15);
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wm * MR + m) * 16 + (lane >> 4) * 4 + r;
                if (co < a.Cout && ci < a.Cin) {
#pragma unroll
                    for (int xi = 0; xi < 16; ++xi) out[(size_t)xi * nwc + (size_t)co * a.Cin + ci] = acc[xi][m][n][r];
                }
            }
    }
    if (a.want_bias && blockIdx.x == 0 && tid < T::COT && co0 + tid < a.Cout) out[16 * nwc + co0 + tid] = bsum;
}

// dg = G^T (sum over splits of dU) G per (co, ci); thread (w, xi) sums position xi of weight w over the splits, the block
// combines the 16 positions through LDS.  G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]].
__global__ __launch_bounds__(1024) void wgrad_wino_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                 float* __restrict__ db, size_t nwc, int Cout, int nsplit) {
    // thread (w, g): weight w of the block's 64, split group g of 16 -- sums all 16 positions over the splits s == g (mod 16)
    // (16 independent accumulators = 16 loads in flight), the block then combines the groups through LDS (two passes of 8
    // positions keep it at 32 KB) and 9 x 64 threads apply G^T . G.
    __shared__ float red[16][8][64];
    __shared__ float u[16][64];
    const int w = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t stride = 16 * nwc + Cout;
    const size_t i = (size_t)blockIdx.x * 64 + w;
    float acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) acc[xi] = 0.f;
    if (i < nwc) {
        for (int s = g; s < nsplit; s += 16) {
            const float* p = partial + (size_t)s * stride + i;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) acc[xi] += p[(size_t)xi * nwc];
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int k = 0; k < 8; ++k) red[g][k][w] = acc[half * 8 + k];
        __syncthreads();
        if (g < 8) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[q][g][w];
            u[half * 8 + g][w] = t;
        }
        __syncthreads();
    }
    if (g < 9 && i < nwc) {
        const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
        const int ky = g / 3, kx = g % 3;
        float t = 0.f;
#pragma unroll
        for (int pa_ = 0; pa_ < 4; ++pa_)
#pragma unroll
            for (int pb_ = 0; pb_ < 4; ++pb_) t += G[pa_][ky] * G[pb_][kx] * u[pa_ * 4 + pb_][w];
        dw[i * 9 + g] = t;
    }
    if (db && blockIdx.x == gridDim.x - 1) {
        // bias: 64 channels per pass, the 16 groups stride over the splits (a serial loop over up to 128 partials per
        // channel was a 40 us chain of dependent loads)
        __syncthreads();
        for (int c0 = 0; c0 < Cout; c0 += 64) {
            const int c = c0 + w;
            float t = 0.f;
            if (c < Cout)
                for (int s = g; s < nsplit; s += 16) t += partial[(size_t)s * stride + 16 * nwc + c];
            red[g][0][w] = t;
            __syncthreads();
            if (g == 0 && c < Cout) {
                float r = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) r += red[q][0][w];
                db[c] = r;
            }
            __syncthreads();
        }
    }
}

// dw[i] (and db) = sum over the split partials.  Block = 64 outputs x 16 split groups; the groups are combined
// through LDS, so even a 1024-element weight with hundreds of partials is a handful of dependent loads per thread.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                            float* __restrict__ db, size_t nw, int Cout, int nsplit) {
    __shared__ float red[16][64];
    const int w = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t n = nw + Cout;
    const size_t i = (size_t)blockIdx.x * 64 + w;
    float v = 0.f;
    if (i < n) {
        int s = g;
        for (; s + 48 < nsplit; s += 64) {   // four independent loads in flight per trip, fixed summation order
            const float p0 = partial[(size_t)s * n + i], p1 = partial[(size_t)(s + 16) * n + i];
            const float p2 = partial[(size_t)(s + 32) * n + i], p3 = partial[(size_t)(s + 48) * n + i];
            v = (((v + p0) + p1) + p2) + p3;
        }
        for (; s < nsplit; s += 16) v += partial[(size_t)s * n + i];
    }
    red[g][w] = v;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][w];
        if (i < nw) dw[i] = t;
        else if (db) db[i - nw] = t;
    }
}

// Weight gradient of the wavelet heads' last convolution (Cout <= 4: an MFMA tile would be > 75 % padding):
//   dW[co,ci,t] = sum_{b,y,x} dz[b,co,y,x] * pad(x)[b,ci,y+ky,x+kx]
// One block per (input channel, image, row slab).  A thread owns one column of one of `nseg` row segments of the
// slab and walks down it with the 3x3 window in registers (3 new loads of x + COUT of dz per pixel), 9*COUT
// accumulators; then a wavefront shuffle + LDS reduction.  Channel-0 blocks also produce the bias partial sums.
// Partials use the same [split][Cout*Cin*9 + Cout] layout as the MFMA path, so wgrad_reduce_kernel finishes both.
template <int COUT>
__global__ __launch_bounds__(256) void conv_wgrad_smallco_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                 float* __restrict__ partial, int C, int H, int W,
                                                                 int pad_mode, int spi, int nseg, int want_bias) {
    const int ci = blockIdx.x, split = blockIdx.y;
    const int b = split / spi, sl = split - b * spi;
    const int rps = (H + spi - 1) / spi;
    const int r0 = sl * rps, r1 = min(H, r0 + rps);
    const int rseg = (max(r1 - r0, 0) + nseg - 1) / nseg;
    const size_t plane = (size_t)H * W;
    const float* xp = x + ((size_t)b * C + ci) * plane;
    const float* gp = dz + (size_t)b * COUT * plane;
    float acc[COUT][9], bs[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        bs[co] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[co][t] = 0.f;
    }
    for (int item = threadIdx.x; item < W * nseg; item += 256) {
        const int seg = item / W, xx = item - seg * W;
        const int y0 = r0 + seg * rseg, y1 = min(r1, y0 + rseg);
        if (y0 >= y1) continue;
        int cx[3];
        bool okx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            cx[d] = xx + d - 1;
            okx[d] = pad_coord(cx[d], W, pad_mode);
            cx[d] = min(max(cx[d], 0), W - 1);
        }
        auto load_row = [&](int yy, float* out) {
            const bool oky = pad_coord(yy, H, pad_mode);
            const float* rp = xp + (size_t)min(max(yy, 0), H - 1) * W;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float v = rp[cx[d]];
                out[d] = (oky && okx[d]) ? v : 0.f;
            }
        };
        float win[9];
        load_row(y0 - 1, win);
        load_row(y0, win + 3);
#pragma unroll 4
        for (int y = y0; y < y1; ++y) {   // unrolled: the next rows' loads are issued ahead of this row's FMAs
            load_row(y + 1, win + 6);
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float g = gp[co * plane + (size_t)y * W + xx];
                bs[co] += g;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[co][t] = fmaf(g, win[t], acc[co][t]);
            }
#pragma unroll
            for (int t = 0; t < 6; ++t) win[t] = win[t + 3];
        }
    }
    __shared__ float red[4][COUT * 10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            float s = t < 9 ? acc[co][t] : bs[co];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) red[wave][co * 10 + t] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < COUT * 10) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const int co = threadIdx.x / 10, t = threadIdx.x % 10;
        const size_t nw = (size_t)COUT * C * 9;
        float* out = partial + (size_t)split * (nw + COUT);
        if (t < 9) out[((size_t)co * C + ci) * 9 + t] = s;
        else if (want_bias && ci == 0) out[nw + co] = s;
    }
}

struct WgradCfg {
    int TH, TW, MR, NC, WM, WN, TAPS;
    void (*launch)(const WgradKArgs&, dim3, hipStream_t);
    const char* name;
};

template <int TH, int TW, int MR, int NC, int WM, int WN, int TAPS>
static void launch_wgrad(const WgradKArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((conv_wgrad_kernel<TH, TW, MR, NC, WM, WN, TAPS>), grid, dim3(WM * WN * 64), 0, s, a);
}
#define WMD_WCFG(TH, TW, MR, NC, WM, WN, TAPS)                                       \
    WgradCfg {                                                                       \
        TH, TW, MR, NC, WM, WN, TAPS, &launch_wgrad<TH, TW, MR, NC, WM, WN, TAPS>,   \
            "conv_wgrad_kernel<" #TH "," #TW "," #MR "," #NC "," #WM "," #WN "," #TAPS ">" \
    }

static const WgradCfg kWCfgs[] = {
    WMD_WCFG(2, 32, 1, 1, 4, 1, 9),   // co64 x ci16, 64-pixel tiles
    WMD_WCFG(2, 32, 1, 1, 2, 2, 9),   // co32 x ci32
    WMD_WCFG(1, 40, 1, 1, 4, 1, 9),   // 40-wide rows: one row per tile keeps the double buffer at 50 KB (3 blocks / CU;
    WMD_WCFG(1, 40, 1, 1, 2, 2, 9),   //   the 2 x 40 tile needs 91 KB = 1 block / CU and ran at 50 instead of 76 TFLOP/s)
    WMD_WCFG(2, 20, 1, 1, 4, 1, 9),   // 20-wide rows (coarsest 640-wide level, NYUv2 15x20)
    WMD_WCFG(2, 20, 1, 1, 2, 2, 9),
    WMD_WCFG(1, 32, 2, 1, 2, 2, 9),   // co64 x ci32, MR = 2: 18 MFMAs per 2 + 9 fragment reads (67 KB: 2 blocks / CU)
    WMD_WCFG(2, 32, 2, 1, 2, 2, 9),   // co64 x ci32, 64-pixel tiles
    WMD_WCFG(1, 32, 2, 2, 2, 1, 9),   // co64 x ci32 in 2 waves: 36 MFMAs per 4 + 18 reads
    WMD_WCFG(1, 40, 2, 1, 2, 2, 9),   // 40-wide rows
    WMD_WCFG(1, 32, 1, 1, 1, 4, 9),   // co16 x ci64: the heads' Cout <= 4 filters on the matrix pipe (3/16 of the rows used)
    WMD_WCFG(1, 40, 1, 1, 1, 4, 9),
    WMD_WCFG(1, 64, 1, 4, 4, 1, 1),   // 1x1: co64 x ci64 over 64 flattened pixels
    WMD_WCFG(1, 64, 1, 2, 2, 2, 1),   // 1x1: co32 x ci64
};
constexpr int kNumWCfgs = sizeof(kWCfgs) / sizeof(kWCfgs[0]);

struct WgradPlan {
    const WgradCfg* cfg;
    int H, W, tiles_x, tiles_y, ntiles, nsplit;
    dim3 grid;
};

static bool plan_wgrad(const wmd_conv_wgrad_args* g, WgradPlan* p) {
    const int taps = g->ksize == 3 ? 9 : 1;
    const int Cin = g->C1 + g->C2;
    const int H = taps == 9 ? g->H : 1, W = taps == 9 ? g->W : g->H * g->W;
    double best = 1e300;
    bool found = false;
    // development: WMD_WGRAD_CFG=<1-based table index> forces a tile, WMD_WGRAD_NSPLIT=<n> the pixel split (read per call)
    const char* e_cfg = getenv("WMD_WGRAD_CFG");
    const char* e_ns = getenv("WMD_WGRAD_NSPLIT");
    const int force = e_cfg ? atoi(e_cfg) - 1 : -1;
    for (int i = 0; i < kNumWCfgs; ++i) {
        const WgradCfg& c = kWCfgs[i];
        if (c.TAPS != taps) continue;
        if (force >= 0 && force < kNumWCfgs && kWCfgs[force].TAPS == taps && i != force) continue;
        if (force < 0 && c.WM * c.MR == 1 && g->Cout > 16) continue;   // 16-row out-channel tiles are for the heads only
        const int TH = c.TH, TW = c.TW;
        const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
        const int cot = c.WM * c.MR * 16, cit = c.WN * c.NC * 16;
        const int gx = (Cin + cit - 1) / cit, gy = (g->Cout + cot - 1) / cot;
        const long ntiles = (long)g->B * tx * ty;
        // padded MACs: every block sweeps all pixel tiles of its split
        const double waste = ((double)gx * cit / Cin) * ((double)gy * cot / g->Cout) * ((double)tx * TW * ty * TH / ((double)H * W));
        // ~2 blocks per CU in flight, at least 4 pixel tiles per block (prologue amortisation), at most 256 partials
        long nsplit = std::max<long>(1, (2L * kNumCU + (long)gx * gy - 1) / ((long)gx * gy));
        nsplit = std::min<long>(nsplit, std::max<long>(1, ntiles / 4));
        nsplit = std::min<long>(nsplit, 256);
        if (e_ns && atoi(e_ns) > 0) nsplit = std::min<long>(atoi(e_ns), std::max<long>(1, ntiles));
        if (g->tune_nsplit > 0) nsplit = std::min<long>(g->tune_nsplit, std::max<long>(1, ntiles));
        const double rounds = std::ceil((double)gx * gy * nsplit / (2.0 * kNumCU));
        const double cost = waste * rounds * 2.0 * kNumCU / ((double)gx * gy * nsplit);
        if (cost < best) {
            best = cost;
            found = true;
            p->cfg = &c;
            p->H = H;
            p->W = W;
            p->tiles_x = tx;
            p->tiles_y = ty;
            p->ntiles = (int)ntiles;
            p->nsplit = (int)nsplit;
            p->grid = dim3((unsigned)gx, (unsigned)gy, (unsigned)nsplit);
        }
    }
    return found;
}

struct WgradWinoCfg {
    int TH, TW, MR, NC, WM, WN;
    void (*launch)(const WgradKArgs&, dim3, hipStream_t);
    const char* name;
    int bpc;      // blocks that share a CU (the pixel split aims at this many rounds-free blocks per CU)
    bool w32;     // the 32x32x2 family (conv_wgrad_wino32_kernel): ~2x the rate of the 16x16x4 form in the cost model
};
template <int TH, int TW, int MR, int NC, int WM, int WN>
static void launch_wgrad_wino(const WgradKArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((conv_wgrad_wino_kernel<TH, TW, MR, NC, WM, WN>), grid, dim3(WM * WN * 64), 0, s, a);
}
#define WMD_WWCFG(TH, TW, MR, NC, WM, WN)                                                  \
    WgradWinoCfg {                                                                         \
        TH, TW, MR, NC, WM, WN, &launch_wgrad_wino<TH, TW, MR, NC, WM, WN>,                \
            "conv_wgrad_wino_kernel<" #TH "," #TW "," #MR "," #NC "," #WM "," #WN ">", 2, false \
    }
// conv_wgrad_wino32_kernel (wmd_conv_wgrad32.hip): WCO x WCI slabs of 32 channels; the table's tile arithmetic sees them as
// MR = NC = 2 sixteen-channel tiles per "wave row / column"
#define WMD_WG32_INST(TH, TW, WCO, WCI)                                                                           \
    WgradWinoCfg{TH, TW, 2, 2, WCO, WCI, &launch_wgrad_wino32<TH, TW, WCO, WCI>,                                  \
                 "conv_wgrad_wino32_kernel<" #TH "," #TW "," #WCO "," #WCI ">", (WCO) * (WCI) >= 4 ? 1 : 2, true},
static const WgradWinoCfg kWWCfgs[] = {
    WMD_WWCFG(2, 32, 1, 1, 4, 1),   // co64 x ci16, 64-pixel tiles (55 KB: 2 blocks / CU)
    WMD_WWCFG(2, 32, 1, 1, 2, 2),   // co32 x ci32
    WMD_WWCFG(2, 32, 1, 2, 4, 1),   // co64 x ci32: two ci tiles per wave share the dz transform (75 KB)
    WMD_WWCFG(2, 32, 2, 1, 2, 2),   // co64 x ci32 in 4 waves
    WMD_WWCFG(2, 32, 1, 2, 2, 1),   // co32 x ci32 in 2 waves
    WMD_WWCFG(2, 40, 1, 1, 4, 1),   // 40-wide rows (W = 40 / 80 / 160 / 320)
    WMD_WWCFG(2, 40, 1, 1, 2, 2),
    WMD_WWCFG(4, 16, 1, 1, 4, 1),   // 16-wide tiles: narrow maps
    WMD_WWCFG(4, 16, 1, 2, 4, 1),
    WMD_WWCFG(2, 32, 1, 1, 1, 4),   // co16 x ci64: the heads' Cout <= 4 filters (3/16 of the MFMA rows carry data)
    WMD_WWCFG(2, 40, 1, 1, 1, 4),
    WMD_WWCFG(2, 32, 1, 2, 1, 2),   // co16 x ci64 in 2 waves
#include "wmd_conv_wgrad32_table.inc"
};
constexpr int kNumWWCfgs = sizeof(kWWCfgs) / sizeof(kWWCfgs[0]);

struct WgradWinoPlan {
    const WgradWinoCfg* cfg;
    int tiles_x, tiles_y, ntiles, nsplit;
    dim3 grid;
};

// Winograd form for the 3x3 layers whose width fills a tile row reasonably; everything else (1x1, 20-wide maps, tiny maps)
// stays on the direct kernel.  WMD_WGRAD_WINO=0 switches the family off, WMD_WGRAD_WINO_CFG=<1-based index> forces a tile.
static bool plan_wgrad_wino(const wmd_conv_wgrad_args* g, WgradWinoPlan* p) {
    if (g->ksize != 3) return false;
    const char* e_on = getenv("WMD_WGRAD_WINO");
    if (e_on && atoi(e_on) == 0) return false;
    if (g->tune_cfg < 0) return false;                       // caller asked for the direct kernel
    const char* e_cfg = getenv("WMD_WGRAD_WINO_CFG");
    const char* e_ns = getenv("WMD_WGRAD_NSPLIT");
    const int force = g->tune_cfg > 0 ? g->tune_cfg - 1 : (e_cfg ? atoi(e_cfg) - 1 : -1);
    if (force >= kNumWWCfgs) return false;
    const int Cin = g->C1 + g->C2, H = g->H, W = g->W;
    double best = 1e300;
    bool found = false;
    for (int i = 0; i < kNumWWCfgs; ++i) {
        if (force >= 0 && force < kNumWWCfgs && i != force) continue;
        const WgradWinoCfg& c = kWWCfgs[i];
        const int tx = (W + c.TW - 1) / c.TW, ty = (H + c.TH - 1) / c.TH;
        const int cot = c.WM * c.MR * 16, cit = c.WN * c.NC * 16;
        if (force < 0 && cot == 16 && g->Cout > 16) continue;   // 16-row tiles are for the heads only
        const int gx = (Cin + cit - 1) / cit, gy = (g->Cout + cot - 1) / cot;
        const long ntiles = (long)g->B * tx * ty;
        const double pix_waste = (double)tx * c.TW * ty * c.TH / ((double)H * W);
        if (force < 0 && pix_waste > 1.6) continue;
        const double waste = ((double)gx * cit / Cin) * ((double)gy * cot / g->Cout) * pix_waste;
        long nsplit = std::max<long>(1, ((long)c.bpc * kNumCU + (long)gx * gy - 1) / ((long)gx * gy));
        nsplit = std::min<long>(nsplit, std::max<long>(1, ntiles / 4));
        static const long ns_cap = [] { const char* e = getenv("WMD_WGRAD_NSPLIT_CAP"); return e && atoi(e) > 0 ? (long)atoi(e) : 256L; }();   // (128 until round 4: single-slab layers -- Cout = 32 -- then ran 256 blocks on 512 slots: L14 242 -> 209 us)
        nsplit = std::min<long>(nsplit, ns_cap);
        if (e_ns && atoi(e_ns) > 0) nsplit = std::min<long>(atoi(e_ns), std::max<long>(1, ntiles));
        if (g->tune_nsplit > 0) nsplit = std::min<long>(g->tune_nsplit, std::max<long>(1, ntiles));
        const double rounds = std::ceil((double)gx * gy * nsplit / ((double)c.bpc * kNumCU));
        // two ci tiles / two co tiles per wave amortise the transforms: small bonus; the 32x32x2 family runs ~2x the rate
        const double eff = c.w32 ? 0.5 : ((c.MR * c.NC > 1) ? 0.93 : 1.0);
        const double cost = eff * waste * rounds * (double)c.bpc * kNumCU / ((double)gx * gy * nsplit);
        if (cost < best) {
            best = cost;
            found = true;
            p->cfg = &c;
            p->tiles_x = tx;
            p->tiles_y = ty;
            p->ntiles = (int)ntiles;
            p->nsplit = (int)nsplit;
            p->grid = dim3((unsigned)gx, (unsigned)gy, (unsigned)nsplit);
        }
    }
    return found;
}

}  // namespace wmd

using namespace wmd;

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int validate_bwd(int B, int H, int W, int C1, int up1, int C2, int Cout, int ksize, int pad_mode, const char* who) {
    if (B <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0)
        return fail(WMD_ERR_BAD_SHAPE, "%s: B=%d H=%d W=%d C1=%d C2=%d Cout=%d", who, B, H, W, C1, C2, Cout);
    if (ksize != 1 && ksize != 3) return fail(WMD_ERR_UNSUPPORTED, "%s: ksize=%d", who, ksize);
    if (up1 != 1 && up1 != 2) return fail(WMD_ERR_BAD_ARG, "%s: up1=%d", who, up1);
    if (up1 == 2 && ((H | W) & 1)) return fail(WMD_ERR_BAD_SHAPE, "%s: up1=2 needs even H,W", who);
    if (pad_mode < 0 || pad_mode > 2) return fail(WMD_ERR_BAD_ARG, "%s: pad_mode=%d", who, pad_mode);
    if (ksize == 3 && pad_mode == WMD_PAD_REFLECT && (H < 2 || W < 2))
        return fail(WMD_ERR_BAD_SHAPE, "%s: reflect padding needs H,W >= 2", who);
    return WMD_OK;
}

static bool dgrad_direct(const wmd_conv_dgrad_args* g) {
    // 1x1 without upsample/concat: the GEMM output IS dx1
    return g->ksize == 1 && g->up1 == 1 && g->C2 == 0 && g->dx1;
}

static void dgrad_conv_args(const wmd_conv_dgrad_args* g, wmd_conv_args* c, float* gbuf, float* ws, size_t ws_floats) {
    const int halo = g->ksize == 3 ? 1 : 0;
    memset(c, 0, sizeof(*c));
    c->B = g->B;
    c->H = g->H + 2 * halo;
    c->W = g->W + 2 * halo;
    c->C1 = g->Cout;  // the reduction runs over the forward's output channels
    c->up1 = 1;
    c->C2 = 0;
    c->Cout = g->C1 + g->C2;  // rows = forward input channels
    c->ksize = g->ksize;
    c->pad_mode = WMD_PAD_ZERO;
    c->act = WMD_ACT_NONE;
    c->x1 = g->dz;
    c->wp = g->wp_dgrad;
    c->wp_wino = g->ksize == 3 ? g->wp_dgrad_wino : nullptr;
    c->y = gbuf;
    c->workspace = ws;
    c->workspace_floats = ws_floats;
    c->tune_cfg = g->tune_cfg;
    c->tune_ksplit = g->tune_ksplit;
    if (dgrad_direct(g) && g->x1_fwd && g->x1_act != WMD_ACT_NONE) {   // the GEMM output IS dx1: gate it in the epilogue
        c->gate = g->x1_fwd;
        c->gate_act = g->x1_act;
        c->gate_slope = g->x1_slope;
    }
}


extern "C" size_t wmd_conv_dgrad_workspace_floats(const wmd_conv_dgrad_args* g) {
    if (!g || g->B <= 0) return 0;
    const int halo = g->ksize == 3 ? 1 : 0;
    const size_t gsz = dgrad_direct(g) ? 0 : (size_t)g->B * (g->C1 + g->C2) * (g->H + 2 * halo) * (g->W + 2 * halo);
    wmd_conv_args c;
    dgrad_conv_args(g, &c, nullptr, nullptr, 0);
    return gsz + wmd_conv_fwd_workspace_floats(&c);
}

extern "C" int wmd_conv_dgrad(const wmd_conv_dgrad_args* g, void* stream) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "wmd_conv_dgrad: null args");
    if (!g->dz || !g->wp_dgrad) return fail(WMD_ERR_BAD_ARG, "wmd_conv_dgrad: null tensor pointer");
    if (!g->dx1 && !g->dx2) return WMD_OK;
    if (g->dx2 && g->C2 <= 0) return fail(WMD_ERR_BAD_ARG, "wmd_conv_dgrad: dx2 given but C2=%d", g->C2);
    int st = validate_bwd(g->B, g->H, g->W, g->C1, g->up1, g->C2, g->Cout, g->ksize, g->pad_mode, "wmd_conv_dgrad");
    if (st) return st;
    if (g->ksize == 1 && g->up1 == 2) return fail(WMD_ERR_UNSUPPORTED, "wmd_conv_dgrad: 1x1 with upsampled input");
    const int halo = g->ksize == 3 ? 1 : 0;
    const size_t gsz = dgrad_direct(g) ? 0 : (size_t)g->B * (g->C1 + g->C2) * (g->H + 2 * halo) * (g->W + 2 * halo);
    if (g->workspace_floats < gsz || (gsz && !g->workspace))
        return fail(WMD_ERR_WORKSPACE, "wmd_conv_dgrad: workspace %zu < %zu floats", g->workspace_floats, gsz);
    float* gbuf = dgrad_direct(g) ? g->dx1 : g->workspace;
    wmd_conv_args c;
    dgrad_conv_args(g, &c, gbuf, g->workspace ? g->workspace + gsz : nullptr, g->workspace_floats - gsz);
    if (c.workspace_floats == 0) c.workspace = nullptr;
    st = run_conv(&c, halo, g->H, g->W, stream);
    if (st || dgrad_direct(g)) return st;
    const size_t n = (g->dx1 ? (size_t)g->B * g->C1 * (g->H / g->up1) * (g->W / g->up1) : 0) +
                     (g->dx2 ? (size_t)g->B * g->C2 * g->H * g->W : 0);
    // a thread folds 4 padded-domain columns when rows split into whole quads (dx pointers come 16-byte aligned
    // from the caller's allocator; the scalar kernel covers everything else)
    const bool vec = g->W % 4 == 0 && ((uintptr_t)g->dx1 % 16 == 0) && ((uintptr_t)g->dx2 % 16 == 0);
    const int work = g->H * g->W / (vec ? 4 : 1);
    const dim3 grid(std::max(1, std::min((work + 255) / 256, 64)), g->B * (g->C1 + g->C2));
    ProfScope prof("conv_dgrad_fold_kernel", (double)n, 4.0 * (gsz + n), (hipStream_t)stream);
    if (vec)
        hipLaunchKernelGGL(conv_dgrad_fold_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, gbuf, g->dx1, g->dx2, g->B, g->C1,
                           g->C2, g->H, g->W, g->up1, g->pad_mode, halo, g->x1_fwd, g->x1_act, g->x1_slope);
    else
        hipLaunchKernelGGL(conv_dgrad_fold_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, gbuf, g->dx1, g->dx2, g->B, g->C1,
                           g->C2, g->H, g->W, g->up1, g->pad_mode, halo, g->x1_fwd, g->x1_act, g->x1_slope);
    return check_launch("conv_dgrad_fold_kernel");
}

static bool smallco_wgrad(const wmd_conv_wgrad_args* g) {
    // WMD_WGRAD_SMALLCO=1: the VALU column-walk kernel for the heads' Cout <= 4 filters (round 1); default: the MFMA
    // kernel with 16-row out-channel tiles (3/16 of the rows used, but the reduction over pixels runs on the matrix pipe
    // and dz is read once per 64 input channels instead of once per channel)
    const char* e = getenv("WMD_WGRAD_SMALLCO");
    if (!(e && atoi(e) == 1)) return false;
    return g->Cout <= 4 && g->ksize == 3 && g->up1 == 1 && g->C2 == 0;
}
struct SmallcoPlan {
    int spi;    // row slabs per image
    int nseg;   // row segments per slab walked concurrently by one block
    int nsplit; // B * spi partial-sum slices
};
static SmallcoPlan smallco_plan(const wmd_conv_wgrad_args* g) {
    SmallcoPlan p;
    // ~2 blocks per CU over (channels x images x slabs) (measured: 1 -> 0.59, 2 -> 0.46, 4 -> 0.53, 8 -> 0.57 ms per step:
    // more slabs = more window prologues and partials), slabs of at least 8 rows
    long spi = (2L * kNumCU + (long)g->C1 * g->B - 1) / ((long)g->C1 * g->B);
    spi = std::max<long>(1, std::min<long>(spi, std::max(1, g->H / 8)));
    p.spi = (int)spi;
    const int rps = (g->H + p.spi - 1) / p.spi;
    // segments: fill the 256 threads with whole columns, but keep the 2-row window prologue amortised
    double best = -1.0;
    p.nseg = 1;
    for (int n = 1; n <= std::min(8, rps); ++n) {
        const int items = g->W * n, passes = (items + 255) / 256, rseg = (rps + n - 1) / n;
        const double score = (double)items / (passes * 256.0) * rseg / (rseg + 2.0);
        if (score > best) { best = score; p.nseg = n; }
    }
    p.nsplit = g->B * p.spi;
    return p;
}

extern "C" int wmd_conv_wgrad_num_configs(void) { return kNumWWCfgs; }
extern "C" const char* wmd_conv_wgrad_config_name(int index) { return (index >= 0 && index < kNumWWCfgs) ? kWWCfgs[index].name : ""; }

extern "C" size_t wmd_conv_wgrad_workspace_floats(const wmd_conv_wgrad_args* g) {
    if (!g || g->B <= 0 || g->Cout <= 0) return 0;
    if (smallco_wgrad(g)) return (size_t)smallco_plan(g).nsplit * ((size_t)g->Cout * g->C1 * 9 + g->Cout);
    WgradWinoPlan wp;
    if (plan_wgrad_wino(g, &wp)) return (size_t)wp.nsplit * (16 * (size_t)g->Cout * (g->C1 + g->C2) + g->Cout);
    WgradPlan p;
    if (!plan_wgrad(g, &p)) return 0;
    const int taps = g->ksize == 3 ? 9 : 1;
    return (size_t)p.nsplit * ((size_t)g->Cout * (g->C1 + g->C2) * taps + g->Cout);
}

extern "C" int wmd_conv_wgrad(const wmd_conv_wgrad_args* g, void* stream) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "wmd_conv_wgrad: null args");
    if (!g->x1 || !g->dz || !g->dw) return fail(WMD_ERR_BAD_ARG, "wmd_conv_wgrad: null tensor pointer");
    if (g->C2 > 0 && !g->x2) return fail(WMD_ERR_BAD_ARG, "wmd_conv_wgrad: C2=%d but x2 is null", g->C2);
    int st = validate_bwd(g->B, g->H, g->W, g->C1, g->up1, g->C2, g->Cout, g->ksize, g->pad_mode, "wmd_conv_wgrad");
    if (st) return st;
    if (g->ksize == 1 && g->up1 == 2) return fail(WMD_ERR_UNSUPPORTED, "wmd_conv_wgrad: 1x1 with upsampled input");
    if ((double)std::max(g->C1, std::max(g->C2, g->Cout)) * g->H * g->W * 4 > 2147483647.0)
        return fail(WMD_ERR_UNSUPPORTED, "wmd_conv_wgrad: a per-image tensor slice exceeds 2 GiB");
    if (smallco_wgrad(g)) {
        const size_t nw = (size_t)g->Cout * g->C1 * 9;
        const SmallcoPlan sp = smallco_plan(g);
        const int nsplit = sp.nsplit;
        if (!g->workspace || g->workspace_floats < (nw + g->Cout) * nsplit)
            return fail(WMD_ERR_WORKSPACE, "wmd_conv_wgrad: workspace %zu < %zu floats", g->workspace_floats, (nw + g->Cout) * nsplit);
        hipStream_t s = (hipStream_t)stream;
        const double pix = (double)g->B * g->H * g->W;
        {
            ProfScope prof("conv_wgrad_smallco_kernel", 18.0 * g->C1 * g->Cout * pix, 4.0 * pix * (g->C1 + g->Cout), s);
            const dim3 grid(g->C1, nsplit);
            switch (g->Cout) {
                case 1: hipLaunchKernelGGL(conv_wgrad_smallco_kernel<1>, grid, dim3(256), 0, s, g->x1, g->dz, g->workspace, g->C1, g->H, g->W, g->pad_mode, sp.spi, sp.nseg, g->dbias != nullptr); break;
                case 2: hipLaunchKernelGGL(conv_wgrad_smallco_kernel<2>, grid, dim3(256), 0, s, g->x1, g->dz, g->workspace, g->C1, g->H, g->W, g->pad_mode, sp.spi, sp.nseg, g->dbias != nullptr); break;
                case 3: hipLaunchKernelGGL(conv_wgrad_smallco_kernel<3>, grid, dim3(256), 0, s, g->x1, g->dz, g->workspace, g->C1, g->H, g->W, g->pad_mode, sp.spi, sp.nseg, g->dbias != nullptr); break;
                default: hipLaunchKernelGGL(conv_wgrad_smallco_kernel<4>, grid, dim3(256), 0, s, g->x1, g->dz, g->workspace, g->C1, g->H, g->W, g->pad_mode, sp.spi, sp.nseg, g->dbias != nullptr); break;
            }
        }
        st = check_launch("conv_wgrad_smallco_kernel");
        if (st) return st;
        ProfScope prof("wgrad_reduce_kernel", (double)nw * nsplit, 4.0 * nw * (nsplit + 1), s);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nw + g->Cout + 63) / 64)), dim3(1024), 0, s, g->workspace, g->dw,
                           g->dbias, nw, g->Cout, nsplit);
        return check_launch("wgrad_reduce_kernel");
    }
    WgradWinoPlan wp;
    if (plan_wgrad_wino(g, &wp)) {
        const int Cin = g->C1 + g->C2;
        const size_t nwc = (size_t)g->Cout * Cin;
        const size_t need = (size_t)wp.nsplit * (16 * nwc + g->Cout);
        if (!g->workspace || g->workspace_floats < need)
            return fail(WMD_ERR_WORKSPACE, "wmd_conv_wgrad: workspace %zu < %zu floats", g->workspace_floats, need);
        WgradKArgs a;
        a.x1 = g->x1;
        a.x2 = g->x2;
        a.dz = g->dz;
        a.partial = g->workspace;
        a.B = g->B;
        a.H = g->H;
        a.W = g->W;
        a.up1 = g->up1;
        a.H1 = g->H / g->up1;
        a.W1 = g->W / g->up1;
        a.C1 = g->C1;
        a.C2 = g->C2;
        a.Cin = Cin;
        a.Cout = g->Cout;
        a.pad_mode = g->pad_mode;
        a.tiles_x = wp.tiles_x;
        a.tiles_y = wp.tiles_y;
        a.ntiles = wp.ntiles;
        a.nsplit = wp.nsplit;
        a.want_bias = g->dbias != nullptr;
        hipStream_t s = (hipStream_t)stream;
        const double pix = (double)g->B * g->H * g->W;
        {
            ProfScope prof(wp.cfg->name, 2.0 * Cin * 9 * g->Cout * pix, 4.0 * (pix * (Cin + g->Cout) + 9.0 * nwc), s);
            wp.cfg->launch(a, wp.grid, s);
        }
        st = check_launch("conv_wgrad_wino_kernel");
        if (st) return st;
        ProfScope prof("wgrad_wino_reduce_kernel", 16.0 * nwc * wp.nsplit, 4.0 * 16 * nwc * (wp.nsplit + 1), s);
        hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3((unsigned)((nwc + 63) / 64)), dim3(1024), 0, s, g->workspace, g->dw, g->dbias,
                           nwc, g->Cout, wp.nsplit);
        return check_launch("wgrad_wino_reduce_kernel");
    }
    WgradPlan p;
    if (!plan_wgrad(g, &p)) return fail(WMD_ERR_UNSUPPORTED, "wmd_conv_wgrad: no kernel configuration");
    const int taps = g->ksize == 3 ? 9 : 1;
    const int Cin = g->C1 + g->C2;
    const size_t nw = (size_t)g->Cout * Cin * taps;
    if (!g->workspace || g->workspace_floats < (nw + g->Cout) * p.nsplit)
        return fail(WMD_ERR_WORKSPACE, "wmd_conv_wgrad: workspace %zu < %zu floats", g->workspace_floats, (nw + g->Cout) * p.nsplit);
    WgradKArgs a;
    a.x1 = g->x1;
    a.x2 = g->x2;
    a.dz = g->dz;
    a.partial = g->workspace;
    a.B = g->B;
    a.H = p.H;
    a.W = p.W;
    a.up1 = taps == 9 ? g->up1 : 1;
    a.H1 = p.H / a.up1;
    a.W1 = p.W / a.up1;
    a.C1 = g->C1;
    a.C2 = g->C2;
    a.Cin = Cin;
    a.Cout = g->Cout;
    a.pad_mode = g->pad_mode;
    a.tiles_x = p.tiles_x;
    a.tiles_y = p.tiles_y;
    a.ntiles = p.ntiles;
    a.nsplit = p.nsplit;
    a.want_bias = g->dbias != nullptr;
    hipStream_t s = (hipStream_t)stream;
    const double pix = (double)g->B * g->H * g->W;
    {
        ProfScope prof(p.cfg->name, 2.0 * Cin * taps * g->Cout * pix, 4.0 * (pix * (Cin + g->Cout) + (double)nw), s);
        p.cfg->launch(a, p.grid, s);
    }
    st = check_launch("conv_wgrad_kernel");
    if (st) return st;
    {
        ProfScope prof("wgrad_reduce_kernel", (double)nw * p.nsplit, 4.0 * nw * (p.nsplit + 1), s);
        const int blocks = (int)((nw + g->Cout + 63) / 64);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(1024), 0, s, g->workspace, g->dw, g->dbias, nw, g->Cout,
                           p.nsplit);
    }
    st = check_launch("wgrad_reduce_kernel");
    return st;
}

// ================================================================================================
// Depthwise 3x3 + ReLU: the first half of the NYUv2 decoders' optional `is_depthwise` Conv3x3
// (NYUv2/networks/layers.py:23-25,70-75: Conv2d(C, C, 3, groups=C, bias=False) -> ReLU, then a bias-free 1x1 which
// runs through the regular MFMA 1x1 convolution).  Same virtual input as the dense kernels: nearest-upsampled x1
// concatenated with x2, padded by 1 in the layer's padding mode -- never materialised.  HBM-bound VALU stencils.
// ================================================================================================
namespace wmd {

__device__ __forceinline__ float dw_input(const float* __restrict__ x1, const float* __restrict__ x2, int b, int c, int gy, int gx,
                                          int C1, int C2, int H, int W, int up1, int pad_mode) {
    const bool ok = pad_coord(gy, H, pad_mode) & pad_coord(gx, W, pad_mode);
    gy = min(max(gy, 0), H - 1);
    gx = min(max(gx, 0), W - 1);
    float v;
    if (c < C1) {
        const int h1 = H / up1, w1 = W / up1;
        v = x1[(((size_t)b * C1 + c) * h1 + gy / up1) * w1 + gx / up1];
    } else {
        v = x2[(((size_t)b * C2 + (c - C1)) * H + gy) * W + gx];
    }
    return ok ? v : 0.f;
}

__global__ void dwconv_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ w,
                                  float* __restrict__ y, int B, int C1, int C2, int H, int W, int up1, int pad_mode) {
    const int C = C1 + C2, plane = H * W;
    const int bc = blockIdx.y, b = bc / C, c = bc % C;
    float wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = w[c * 9 + t];
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += gridDim.x * blockDim.x) {
        const int yy = p / W, xx = p - yy * W;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) s = fmaf(wt[t], dw_input(x1, x2, b, c, yy + t / 3 - 1, xx + t % 3 - 1, C1, C2, H, W, up1, pad_mode), s);
        y[(size_t)bc * plane + p] = fmaxf(s, 0.f);
    }
}

// padded-domain gradient gP[b,c,Y,X] (Y in [0,H+2), X in [0,W+2)) = sum_t w[c,t] dz[Y-ky, X-kx],  dz = dy * (y > 0);
// conv_dgrad_fold_kernel then applies the adjoint of pad + concat + upsample
__global__ void dwconv_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ w,
                                       float* __restrict__ gp, int C, int H, int W) {
    const int bc = blockIdx.y, c = bc % C;
    const int Hp = H + 2, Wp = W + 2, pplane = Hp * Wp;
    float wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = w[c * 9 + t];
    const float* dyp = dy + (size_t)bc * H * W;
    const float* yp = y + (size_t)bc * H * W;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < pplane; p += gridDim.x * blockDim.x) {
        const int Y = p / Wp, X = p - Y * Wp;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int oy = Y - t / 3, ox = X - t % 3;   // output pixel whose tap t reads padded position (Y, X)
            if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
                const int q = oy * W + ox;
                s = fmaf(wt[t], yp[q] > 0.f ? dyp[q] : 0.f, s);
            }
        }
        gp[(size_t)bc * pplane + p] = s;
    }
}

// dW[c,t] = sum_{b,y,x} dz * P(b,c,y+ky-1,x+kx-1): one block per channel, threads stride over (b, pixel)
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                                const float* __restrict__ dy, const float* __restrict__ y,
                                                                float* __restrict__ dw, int B, int C1, int C2, int H, int W,
                                                                int up1, int pad_mode) {
    const int C = C1 + C2, plane = H * W, c = blockIdx.x;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int i = threadIdx.x; i < B * plane; i += blockDim.x) {
        const int b = i / plane, p = i - b * plane, yy = p / W, xx = p - yy * W;
        const size_t o = ((size_t)b * C + c) * plane + p;
        const float g = y[o] > 0.f ? dy[o] : 0.f;
        if (g == 0.f) continue;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = fmaf(g, dw_input(x1, x2, b, c, yy + t / 3 - 1, xx + t % 3 - 1, C1, C2, H, W, up1, pad_mode), acc[t]);
    }
    __shared__ float red[4][9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float s = acc[t];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = s;
    }
    __syncthreads();
    if (threadIdx.x < 9) dw[c * 9 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

}  // namespace wmd

static int dw_check(const char* who, const wmd_dwconv_args* g) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "%s: null args", who);
    if (!g->x1 || !g->w) return fail(WMD_ERR_BAD_ARG, "%s: null tensor pointer", who);
    if (g->C2 > 0 && !g->x2) return fail(WMD_ERR_BAD_ARG, "%s: C2=%d but x2 is null", who, g->C2);
    int st = validate_bwd(g->B, g->H, g->W, g->C1, g->up1, g->C2, 1, 3, g->pad_mode, who);
    if (st) return st;
    if ((double)(g->C1 + g->C2) * g->B > 65535.0) return fail(WMD_ERR_UNSUPPORTED, "%s: B*C > 65535", who);
    if ((double)g->B * (g->C1 + g->C2) * (g->H + 2) * (g->W + 2) > 2147483647.0) return fail(WMD_ERR_UNSUPPORTED, "%s: more than 2^31 elements", who);
    return WMD_OK;
}

extern "C" int wmd_dwconv3x3_fwd(const wmd_dwconv_args* g, float* y, void* stream) {
    int st = dw_check("wmd_dwconv3x3_fwd", g);
    if (st) return st;
    if (!y) return fail(WMD_ERR_BAD_ARG, "wmd_dwconv3x3_fwd: null output");
    const int C = g->C1 + g->C2, plane = g->H * g->W;
    hipStream_t s = (hipStream_t)stream;
    const double n = (double)g->B * C * plane;
    ProfScope prof("dwconv_fwd_kernel", 18.0 * n, 8.0 * n, s);
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(std::max(1, std::min((plane + 255) / 256, 64)), g->B * C), dim3(256), 0, s, g->x1, g->x2, g->w, y,
                       g->B, g->C1, g->C2, g->H, g->W, g->up1, g->pad_mode);
    return check_launch("dwconv_fwd_kernel");
}

extern "C" size_t wmd_dwconv3x3_bwd_workspace_floats(const wmd_dwconv_args* g) {
    return g ? (size_t)g->B * (g->C1 + g->C2) * (g->H + 2) * (g->W + 2) : 0;
}

extern "C" int wmd_dwconv3x3_bwd(const wmd_dwconv_args* g, const float* y, const float* dy, float* dx1, float* dx2, float* dw,
                                 float* workspace, size_t workspace_floats, void* stream) {
    int st = dw_check("wmd_dwconv3x3_bwd", g);
    if (st) return st;
    if (!y || !dy) return fail(WMD_ERR_BAD_ARG, "wmd_dwconv3x3_bwd: null tensor pointer");
    if (dx2 && g->C2 <= 0) return fail(WMD_ERR_BAD_ARG, "wmd_dwconv3x3_bwd: dx2 given but C2=%d", g->C2);
    const int C = g->C1 + g->C2, plane = g->H * g->W, pplane = (g->H + 2) * (g->W + 2);
    hipStream_t s = (hipStream_t)stream;
    const double n = (double)g->B * C * plane;
    if (dx1 || dx2) {
        if (!workspace || workspace_floats < wmd_dwconv3x3_bwd_workspace_floats(g))
            return fail(WMD_ERR_WORKSPACE, "wmd_dwconv3x3_bwd: workspace %zu < %zu floats", workspace_floats, wmd_dwconv3x3_bwd_workspace_floats(g));
        {
            ProfScope prof("dwconv_bwd_data_kernel", 18.0 * n, 12.0 * n, s);
            hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(std::max(1, std::min((pplane + 255) / 256, 64)), g->B * C), dim3(256), 0, s, dy, y,
                               g->w, workspace, C, g->H, g->W);
        }
        st = check_launch("dwconv_bwd_data_kernel");
        if (st) return st;
        const bool vec = g->W % 4 == 0 && ((uintptr_t)dx1 % 16 == 0) && ((uintptr_t)dx2 % 16 == 0);
        const int work = plane / (vec ? 4 : 1);
        const dim3 grid(std::max(1, std::min((work + 255) / 256, 64)), g->B * C);
        ProfScope prof("conv_dgrad_fold_kernel", n, 8.0 * n, s);
        if (vec)
            hipLaunchKernelGGL(conv_dgrad_fold_kernel<true>, grid, dim3(256), 0, s, workspace, dx1, dx2, g->B, g->C1, g->C2, g->H, g->W, g->up1,
                               g->pad_mode, 1, (const float*)nullptr, 0, 0.f);
        else
            hipLaunchKernelGGL(conv_dgrad_fold_kernel<false>, grid, dim3(256), 0, s, workspace, dx1, dx2, g->B, g->C1, g->C2, g->H, g->W, g->up1,
                               g->pad_mode, 1, (const float*)nullptr, 0, 0.f);
        st = check_launch("conv_dgrad_fold_kernel");
        if (st) return st;
    }
    if (dw) {
        ProfScope prof("dwconv_bwd_weight_kernel", 18.0 * n, 12.0 * n, s);
        hipLaunchKernelGGL(dwconv_bwd_weight_kernel, dim3(C), dim3(256), 0, s, g->x1, g->x2, dy, y, dw, g->B, g->C1, g->C2, g->H, g->W, g->up1,
                           g->pad_mode);
        st = check_launch("dwconv_bwd_weight_kernel");
    }
    return st;
}
