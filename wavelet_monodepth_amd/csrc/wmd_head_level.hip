// One launch per pyramid level for the inference form of the two high-frequency wavelet heads
// (KITTI/networks/decoders/depth_decoder.py:108-136,164-166):
//
//   mid_s = LeakyReLU(W1_s x + b1_s)                              s in {+,-},  1x1, C -> C
//   h_s   = b3_s + Conv3x3_reflect(mid_s; W3_s)                   C -> 3
//   yh    = 2^(s-1) sigmoid(h_+) - 2^(s-1) sigmoid(h_-)
//   out   = HaarIDWT(yl, yh);  disp = clamp(out * disp_scale, 0, 1)
//
// A block owns a 4 x 40 tile of coefficient pixels and everything it needs stays on chip:
//   1. the (4+2) x (40+2) halo patch of x, all C channels, is gathered into LDS by LDS-DMA (reflect / replicate /
//      zero padding resolved in the per-lane gather offsets, exactly like the trunk convolution);
//   2. GEMM 1 (MFMA 16x16x4 f32), both sides in one pass over the patch:  mid_s[C x 256 patch positions] = W1_s x,
//      bias + LeakyReLU, written to LDS (the - side over the patch, which is dead by then).  mid at a halo position
//      is mid at the padded coordinate, which is what the padded 3x3 of the reference reads;
//   3. GEMM 2:  t_s[27 x 256] = W3'_s mid_s  -- the 3x3 regrouped as 27 "tap-partial" 1x1 outputs (row co*9+tap),
//      4.5x fewer MFMAs than nine shifted 3 -> 16-padded products; t_s replaces mid_s in LDS;
//   4. every pixel sums its nine shifted tap-partials (2 x 27 ds_reads), + bias, sigmoid, combine, Haar butterfly,
//      stores.  Five barriers per block in total; the weight fragments (64 VGPRs) are fetched behind the patch DMA.
// The two-launch form (wmd_head_fused_fwd + wmd_head_shiftsum_fwd) writes and re-reads 54 planes per level
// (212 MB per config-2 step); this one reads x once (halo re-reads hit L2) and writes only the results.
// MFMA-bound by construction: (2 C^2 + 54 C) MACs per position against C*4 bytes read.
#include <algorithm>
#include "wmd_internal.h"

namespace wmd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void hl_dma4(__amdgpu_buffer_rsrc_t r, lds_ptr_t dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 4, voff, soff, 0, 0);
}

constexpr int HL_TH = 4, HL_TW = 40, HL_PH = HL_TH + 2, HL_PW = HL_TW + 2;
constexpr int HL_NPOS = HL_PH * HL_PW;   // 252 patch positions
constexpr int HL_PS = 272;               // per-channel LDS stride: >= 256 MFMA columns (16 groups of 16) and == 16 (mod 32)
constexpr int HL_NPIX = HL_TH * HL_TW;   // 160

// mid / tap-partial planes are stored with the position index XOR-swizzled by the row: a ds_write_b128 is serviced
// in groups of 8 consecutive lanes = 8 consecutive rows at one position quad, and with the row stride == 16 banks
// (what the MFMA operand reads want) rows r, r+2, r+4, r+6 would hit the same four banks (4-way conflict, measured as
// a third of the kernel).  XOR-ing the position with ((row >> 1) & 3) * 4 spreads them over all 32 banks; an operand
// read touches rows 4kk .. 4kk+3 whose swizzle is the same within each 32-lane service group, so it stays conflict-free.
__device__ __forceinline__ constexpr int hl_swz(int row) { return ((row >> 1) & 3) * 4; }

// NW waves; wave w owns position groups [w*NR, (w+1)*NR), NR = 16 / NW, and all row tiles of both sides.
// LDS: the patch (C x 272 floats) + one more such plane; mid and then the tap-partials of the + side live in the
// second plane, those of the - side replace the patch once GEMM 1 has consumed it.  C = 32: 70 KB (2 blocks / CU).
template <int C, int NW>
__global__ __launch_bounds__(NW * 64, 2) void head_level_kernel(const wmd_head_level_args a, int tiles_x, int tiles_y, int ntiles) {
    constexpr int MR = C / 16, NR = 16 / NW, KS = C / 4, PS = HL_PS, PW = HL_PW;
    static_assert(C % 16 == 0 && C >= 32, "mid rows must cover the 32 tap-partial rows that replace them");
    static_assert(16 % NW == 0, "whole position groups per wave");
    __shared__ __attribute__((aligned(16))) float xs[C * PS];
    __shared__ __attribute__((aligned(16))) float ms[C * PS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const size_t plane = (size_t)H * W;

    // weight fragments of both sides, straight from the packed images (L2-resident, the same for every block);
    // loaded once per block: blocks are persistent (a grid of 2 per CU strides over the tiles)
    float w1f[2][MR][KS], w2f[2][2][KS];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const float* w1 = a.wp1 + (size_t)side * MR * KS * 64 + lane;
        const float* w2 = a.wp2 + (size_t)side * 2 * KS * 64 + lane;
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int k = 0; k < KS; ++k) w1f[side][m][k] = w1[(m * KS + k) * 64];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < KS; ++k) w2f[side][j][k] = w2[(j * KS + k) * 64];
    }
    // biases: also once per block (a dependent global load in the middle of a tile is ~1-2 us of exposed latency
    // with only two blocks per CU to hide it)
    float b1v[2][MR], b3v[2][3];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
#pragma unroll
        for (int m = 0; m < MR; ++m) b1v[sd][m] = a.bias1 ? a.bias1[sd * C + m * 16 + (lane & 15)] : 0.f;
        const float* b3 = sd == 0 ? a.bias_p : a.bias_n;
#pragma unroll
        for (int co = 0; co < 3; ++co) b3v[sd][co] = b3 ? b3[co] : 0.f;
    }
    // tile sequence dealt so that an XCD (block id mod 8; gridDim.x is a multiple of 8) owns one contiguous run of it: the halo
    // rows / columns of neighbouring tiles are fetched into ONE L2 (see head3x3_kernel)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * HL_TH, x0 = tx * HL_TW;

    // this thread's low-pass coefficient, requested now and consumed after the last barrier
    const int py = y0 + tid / HL_TW, px = x0 + tid % HL_TW;
    const bool pix_ok = tid < HL_NPIX && py < H && px < W;
    const float yl_v = (pix_ok && a.yl) ? a.yl[(size_t)b * plane + (size_t)py * W + px] : 0.f;

    // Block-sparse levels (yh_mask = the wavelet mask of the sparse decoders): a tile without a single mask pixel has yh = 0
    // everywhere, so its synthesis is the low-pass alone -- no patch, no GEMMs.  (Ballot per wave + flags in LDS; the flags are
    // rewritten only after this tile's barriers.)
    if (a.yh_mask) {
        __shared__ int tile_any[NW];
        const bool mine = pix_ok && a.yh_mask[(size_t)b * plane + (size_t)py * W + px] != 0;
        const bool wave_any = __builtin_amdgcn_ballot_w64(mine) != 0;
        if (lane == 0) tile_any[wave] = wave_any ? 1 : 0;
        __syncthreads();
        int all = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) all |= tile_any[w];
        if (__builtin_amdgcn_readfirstlane(all) == 0) {
            if (pix_ok) {
#pragma unroll
                for (int co = 0; co < 3; ++co) a.yh[((size_t)b * 3 + co) * plane + (size_t)py * W + px] = 0.f;
                if (a.yl && a.out) {
                    float v = yl_v * 0.5f;
                    const size_t dst = (size_t)b * 4 * plane + (size_t)(2 * py) * (2 * W) + 2 * px;
                    *reinterpret_cast<float2*>(a.out + dst) = make_float2(v, v);
                    *reinterpret_cast<float2*>(a.out + dst + 2 * W) = make_float2(v, v);
                    if (a.disp) {
                        v *= a.disp_scale;
                        if (a.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
                        *reinterpret_cast<float2*>(a.disp + dst) = make_float2(v, v);
                        *reinterpret_cast<float2*>(a.disp + dst + 2 * W) = make_float2(v, v);
                    }
                }
            }
            __syncthreads();   // the flags are free for the next tile
            continue;
        }
    }

    // ---- 1. gather the patch: wave w moves positions [(w&3)*64, +64) of channels j = (w>>2), (w>>2)+NW/4, ... ----
    {
        const int p = (wave & 3) * 64 + lane;
        int gy = y0 + p / PW - 1, gx = x0 + p % PW - 1;
        bool ok = p < HL_NPOS;
        ok = pad_coord(gy, H, a.pad_mode) && ok;
        ok = pad_coord(gx, W, a.pad_mode) && ok;
        ok = ok && gy >= 0 && gx >= 0 && gy < H && gx < W;   // tile overhang
        gy = min(max(gy, 0), H - 1);
        gx = min(max(gx, 0), W - 1);
        const unsigned off = ok ? (unsigned)(gy * W + gx) * 4u : 0x80000000u;   // out of range -> the DMA writes 0
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x + (size_t)b * C * plane), 0, (int)(C * plane * 4), 0x00020000);
        constexpr int CSTEP = NW / 4;
#pragma unroll 8
        for (int j = wave >> 2; j < C; j += CSTEP)
            hl_dma4(rx, (lds_ptr_t)(xs + j * PS + (wave & 3) * 64), off, (unsigned)j * (unsigned)(plane * 4));
    }

    __syncthreads();   // the patch has landed (vmcnt(0) precedes the barrier)

    // ---- 2. GEMM 1, both sides at once (every patch fragment feeds 2*MR MFMAs): mid = LeakyReLU(W1 x + b1) ---------
    f32x4 acc[2][MR][NR];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n) acc[sd][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frag = (lane >> 4) * PS + wave * NR * 16 + (lane & 15);   // (k-lane, position) of this lane's B element
    {
        float pf[2][NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) pf[0][n] = xs[frag + n * 16];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k + 1 < KS) {
#pragma unroll
                for (int n = 0; n < NR; ++n) pf[(k + 1) & 1][n] = xs[frag + (k + 1) * 4 * PS + n * 16];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int n = 0; n < NR; ++n)
                        acc[sd][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[k & 1][n], w1f[sd][m][k], acc[sd][m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();   // every wave is done reading the patch: mid of the - side may overwrite it
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
        float* dstm = sd == 0 ? ms : xs;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int ch = m * 16 + (lane & 15);
            const float bv = b1v[sd][m];
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int q = (wave * NR + n) * 16 + (lane >> 4) * 4;
                float4 v;
                v.x = act_apply(acc[sd][m][n][0] + bv, WMD_ACT_LEAKY, a.slope);
                v.y = act_apply(acc[sd][m][n][1] + bv, WMD_ACT_LEAKY, a.slope);
                v.z = act_apply(acc[sd][m][n][2] + bv, WMD_ACT_LEAKY, a.slope);
                v.w = act_apply(acc[sd][m][n][3] + bv, WMD_ACT_LEAKY, a.slope);
                *reinterpret_cast<float4*>(dstm + ch * PS + (q ^ hl_swz(ch))) = v;
            }
        }
    }
    __syncthreads();
    // training forward (wmd_head_level_args.mid_out): the tile's own pixels of mid, both sides, out of LDS -- consecutive
    // threads take consecutive pixels of a channel (40-pixel runs); the halo positions belong to the neighbouring tiles
    if (a.mid_out) {
        const bool v4 = (W & 3) == 0;      // (x0 and rx are multiples of 4: 16-byte stores wherever the image width allows)
        for (int idx = tid; idx < 2 * C * HL_TH * (HL_TW / 4); idx += NW * 64) {
            const int x4 = idx % (HL_TW / 4), r2 = idx / (HL_TW / 4);
            const int ry = r2 % HL_TH, chs = r2 / HL_TH;
            const int sd = chs / C, ch = chs - sd * C;
            const int rx = x4 * 4, q0 = (ry + 1) * PW + rx + 1;
            const float* pl = (sd == 0 ? ms : xs) + ch * PS;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pl[(q0 + e) ^ hl_swz(ch)];
            if (y0 + ry >= H || x0 + rx >= W) continue;
            float* dst = a.mid_out + ((size_t)b * a.mid_ct + (sd == 0 ? a.mid_off_p : a.mid_off_n) + ch) * plane + (size_t)(y0 + ry) * W + x0 + rx;
            if (v4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (x0 + rx + e < W) dst[e] = v[e];
            }
        }
    }

    // ---- 3. GEMM 2, both sides: t = W3' mid -----------------------------------------------------------------------
    f32x4 acc2[2][2][NR];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int n = 0; n < NR; ++n) acc2[sd][j][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
        // rows 4k .. 4k+3 of K-step k: this lane reads row 4k + (lane >> 4), whose swizzle is ((2k + (lane >> 5)) & 3) * 4
        const float* mbase = (sd == 0 ? ms : xs) + (lane >> 4) * PS + wave * NR * 16;
        const int io[2] = {(lane & 15) ^ ((lane >> 5) * 4), (lane & 15) ^ (((2 + (lane >> 5)) & 3) * 4)};
        float pf[2][NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) pf[0][n] = mbase[n * 16 + io[0]];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k + 1 < KS) {
#pragma unroll
                for (int n = 0; n < NR; ++n) pf[(k + 1) & 1][n] = mbase[(k + 1) * 4 * PS + n * 16 + io[(k + 1) & 1]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int n = 0; n < NR; ++n)
                    acc2[sd][j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[k & 1][n], w2f[sd][j][k], acc2[sd][j][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();   // every wave is done reading mid: t may overwrite it
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
        float* dstt = sd == 0 ? ms : xs;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = j * 16 + (lane & 15);   // rows 27..31 are padding (zero weights)
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int q = (wave * NR + n) * 16 + (lane >> 4) * 4;
                *reinterpret_cast<float4*>(dstt + row * PS + (q ^ hl_swz(row))) =
                    make_float4(acc2[sd][j][n][0], acc2[sd][j][n][1], acc2[sd][j][n][2], acc2[sd][j][n][3]);
            }
        }
    }
    __syncthreads();

    // ---- 4. nine-tap shift-sum ----------------------------------------------------------------------------------------
    if (tid < HL_NPIX) {
    const int q_shift = (tid / HL_TW) * PW + tid % HL_TW;   // patch position of this thread's pixel, tap (0,0)
    float hs[2][3];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
        const float* ts = sd == 0 ? ms : xs;
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            float s = b3v[sd][co];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) s += ts[(co * 9 + tp) * PS + ((q_shift + (tp / 3) * PW + tp % 3) ^ hl_swz(co * 9 + tp))];
            hs[sd][co] = s;
        }
    }

    // ---- combine, store, Haar synthesis ----------------------------------------------------------------------------
    const int y = y0 + tid / HL_TW, x = x0 + tid % HL_TW;
    if (y < H && x < W) {
    float yh[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) {
        const float a1 = 1.f / (1.f + expf(-hs[0][co])), a2 = 1.f / (1.f + expf(-hs[1][co]));
        yh[co] = a.scale * a1 - a.scale * a2;
        if (a.sig_p) {   // training forward: what the heads' backward multiplies by
            a.sig_p[((size_t)b * 3 + co) * plane + (size_t)y * W + x] = a1;
            a.sig_n[((size_t)b * 3 + co) * plane + (size_t)y * W + x] = a2;
        }
        if (a.yh_mask && a.yh_mask[(size_t)b * plane + (size_t)y * W + x] == 0) yh[co] = 0.f;
        a.yh[((size_t)b * 3 + co) * plane + (size_t)y * W + x] = yh[co];
    }
    if (a.yl && a.out) {
        const float l = yl_v;
        float v[4] = {(l + yh[0] + yh[1] + yh[2]) * 0.5f, (l + yh[0] - yh[1] - yh[2]) * 0.5f,
                      (l - yh[0] + yh[1] - yh[2]) * 0.5f, (l - yh[0] - yh[1] + yh[2]) * 0.5f};
        const size_t dst = (size_t)b * 4 * plane + (size_t)(2 * y) * (2 * W) + 2 * x;
        *reinterpret_cast<float2*>(a.out + dst) = make_float2(v[0], v[1]);
        *reinterpret_cast<float2*>(a.out + dst + 2 * W) = make_float2(v[2], v[3]);
        if (a.disp) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] *= a.disp_scale;
                if (a.clamp01) v[k] = fminf(fmaxf(v[k], 0.f), 1.f);
            }
            *reinterpret_cast<float2*>(a.disp + dst) = make_float2(v[0], v[1]);
            *reinterpret_cast<float2*>(a.disp + dst + 2 * W) = make_float2(v[2], v[3]);
        }
    }
    }
    }
    __syncthreads();   // the tap-partials are consumed: the next tile's patch may land
    }
}

}  // namespace wmd

using namespace wmd;

extern "C" int wmd_head_level_supported(int C) { return C == 32; }

extern "C" int wmd_head_level_fwd(const wmd_head_level_args* g, void* stream) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "wmd_head_level_fwd: null args");
    if (!g->x || !g->wp1 || !g->wp2 || !g->yh) return fail(WMD_ERR_BAD_ARG, "wmd_head_level_fwd: null tensor pointer");
    if (g->B <= 0 || g->H <= 0 || g->W <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_head_level_fwd: B=%d H=%d W=%d", g->B, g->H, g->W);
    if (!wmd_head_level_supported(g->C))
        return fail(WMD_ERR_UNSUPPORTED, "wmd_head_level_fwd: C=%d (32 only; use wmd_head_fused_fwd + wmd_head_shiftsum_fwd)", g->C);
    if (g->pad_mode < 0 || g->pad_mode > 2) return fail(WMD_ERR_BAD_ARG, "wmd_head_level_fwd: pad_mode=%d", g->pad_mode);
    if (g->pad_mode == WMD_PAD_REFLECT && (g->H < 2 || g->W < 2))
        return fail(WMD_ERR_BAD_SHAPE, "wmd_head_level_fwd: reflect padding needs H,W >= 2");
    if ((g->out != nullptr) != (g->yl != nullptr)) return fail(WMD_ERR_BAD_ARG, "wmd_head_level_fwd: yl and out go together");
    if (g->disp && !g->out) return fail(WMD_ERR_BAD_ARG, "wmd_head_level_fwd: disp needs out");
    if ((double)g->C * g->H * g->W * 4 > 2147483647.0)
        return fail(WMD_ERR_UNSUPPORTED, "wmd_head_level_fwd: a per-image tensor slice exceeds 2 GiB");
    if ((g->sig_p != nullptr) != (g->sig_n != nullptr) || ((g->sig_p || g->mid_out) && g->yh_mask))
        return fail(WMD_ERR_BAD_ARG, "wmd_head_level_fwd: sig_p and sig_n go together; the training outputs take no yh_mask");
    if (g->mid_out && (g->mid_ct <= 0 || g->mid_off_p < 0 || g->mid_off_n < 0 || g->mid_off_p + g->C > g->mid_ct || g->mid_off_n + g->C > g->mid_ct))
        return fail(WMD_ERR_BAD_ARG, "wmd_head_level_fwd: mid_out channel offsets outside mid_ct = %d", g->mid_ct);
    hipStream_t s = (hipStream_t)stream;
    if (head_stream_launch(g, nullptr, 0, s)) return check_launch("head_stream_kernel");   // round 6: plain inference outputs stream (wmd_head_stream.hip)
    const int tiles_x = (g->W + HL_TW - 1) / HL_TW, tiles_y = (g->H + HL_TH - 1) / HL_TH;
    const double pix = (double)g->B * g->H * g->W;
    ProfScope prof("head_level_kernel", 2.0 * pix * (2.0 * g->C * g->C + 54.0 * g->C),
                   4.0 * pix * (g->C + 3 + (g->out ? (g->disp ? 9 : 5) : 0)), s);
    const int ntiles = g->B * tiles_x * tiles_y;
    const dim3 grid((unsigned)std::min(ntiles, 2 * kNumCU));   // persistent: 70 KB of LDS = 2 blocks per CU
    hipLaunchKernelGGL((head_level_kernel<32, 4>), grid, dim3(256), 0, s, *g, tiles_x, tiles_y, ntiles);
    return check_launch("head_level_kernel");
}

// Round 6: the level's heads + synthesis AND the completions of up to three coarser levels in one launch (head_stream_kernel's
// pyramid, wmd_head_stream.hip).  coarse[k] as for wmd_head_shiftsum_chain_fwd (coarse to fine, each twice the one before, the
// first takes exactly one of yl / yl_out, the others the chain's low-pass); the finest coarse level is half this level's size and
// its synthesis output is this level's low-pass input: args->yl must be NULL, args->out is required.
extern "C" int wmd_head_level_pyramid_supported(int C, int B, int H, int W) {   // 1: runs, 2: runs and pays (see head_stream_pyramid_pays)
    if (!(wmd_head_level_supported(C) && (H % 8) == 0 && (W % 8) == 0 && H >= 8 && W >= 8 && B > 0)) return 0;
    return head_stream_pyramid_pays(B, H, W) ? 2 : 1;
}

extern "C" int wmd_head_level_pyramid_fwd(const wmd_head_level_args* g, const wmd_head_shiftsum_args* coarse, int n_coarse, void* stream) {
    if (!g || !coarse || n_coarse < 1 || n_coarse > 3) return fail(WMD_ERR_BAD_ARG, "wmd_head_level_pyramid_fwd: args and 1..3 coarse levels");
    if (!g->x || !g->wp1 || !g->wp2 || !g->yh || !g->out) return fail(WMD_ERR_BAD_ARG, "wmd_head_level_pyramid_fwd: null tensor pointer (x, wp1, wp2, yh, out)");
    if (g->yl) return fail(WMD_ERR_BAD_ARG, "wmd_head_level_pyramid_fwd: the low-pass input comes from the pyramid: yl must be NULL");
    if (g->yh_mask || g->mid_out || g->sig_p || g->sig_n || g->pad_mode != WMD_PAD_REFLECT || !(g->slope >= 0.f && g->slope <= 1.f))
        return fail(WMD_ERR_UNSUPPORTED, "wmd_head_level_pyramid_fwd: plain inference outputs, reflect padding, 0 <= slope <= 1 only");
    if (!wmd_head_level_pyramid_supported(g->C, g->B, g->H, g->W))
        return fail(WMD_ERR_UNSUPPORTED, "wmd_head_level_pyramid_fwd: C=%d %dx%d (C = 32, sizes that are multiples of 8)", g->C, g->H, g->W);
    if ((double)g->C * g->H * g->W * 4 > 2147483647.0) return fail(WMD_ERR_UNSUPPORTED, "wmd_head_level_pyramid_fwd: a per-image tensor slice exceeds 2 GiB");
    for (int k = 0; k < n_coarse; ++k) {
        const wmd_head_shiftsum_args& c = coarse[k];
        const int sh = n_coarse - k;
        if (!c.t || !c.yh || !c.out) return fail(WMD_ERR_BAD_ARG, "wmd_head_level_pyramid_fwd: coarse level %d: t, yh and out are required", k);
        if (c.B != g->B || c.H != (g->H >> sh) || c.W != (g->W >> sh) || (c.H << sh) != g->H || (c.W << sh) != g->W)
            return fail(WMD_ERR_BAD_SHAPE, "wmd_head_level_pyramid_fwd: coarse level %d is %dx%dx%d, expected %dx%dx%d", k, c.B, c.H, c.W, g->B, g->H >> sh, g->W >> sh);
        if (c.pad_mode < 0 || c.pad_mode > 2 || (c.pad_mode == WMD_PAD_REFLECT && (c.H < 2 || c.W < 2)))
            return fail(WMD_ERR_BAD_ARG, "wmd_head_level_pyramid_fwd: coarse level %d: pad_mode=%d on %dx%d", k, c.pad_mode, c.H, c.W);
        if (c.yh_mask || c.range_keys || c.sig_p || c.sig_n || c.sig_ll)
            return fail(WMD_ERR_UNSUPPORTED, "wmd_head_level_pyramid_fwd: dense inference only (no yh_mask / range_keys / sigmoid outputs)");
        if (k == 0 ? ((c.yl != nullptr) == (c.yl_out != nullptr)) : (c.yl != nullptr || c.yl_out != nullptr))
            return fail(WMD_ERR_BAD_ARG, "wmd_head_level_pyramid_fwd: the first coarse level takes exactly one of yl / yl_out, the others the chain's low-pass");
    }
    if (!head_stream_launch(g, coarse, n_coarse, (hipStream_t)stream))
        return fail(WMD_ERR_UNSUPPORTED, "wmd_head_level_pyramid_fwd: the streaming kernel is switched off (WMD_HEAD_STREAM=0)");
    return check_launch("head_stream_kernel");
}
