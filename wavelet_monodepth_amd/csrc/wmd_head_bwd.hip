// Backward of the 3x3 stage of a level's wavelet heads (training path) for gfx950.
//
// Reference: the heads are Conv1x1 -> LeakyReLU -> Conv3x3(C, 3 | 1, reflection padding) -> sigmoid
// (KITTI/networks/decoders/depth_decoder.py:104-136, NYUv2/networks/decoders/densedepth_decoder.py:104-127); their backward
// comes from torch.autograd (KITTI/trainer.py:211, NYUv2/train.py:327).  The generic kernels treat the 2-3 heads of a level as
// one block-diagonal [6|7, Ct, 3, 3] filter: the weight gradient pads 6 output channels to 16 MFMA rows, the data gradient
// pads a 6-deep reduction to 16 and then folds a padded-domain buffer -- 172 + 126 us at the finest level of BASELINE config 2
// for 2.5 GFLOP (profiles/r03_backward_notes.md).  Here the 3x3 is regrouped like the inference heads' tap-partials:
//
//     g[(o,tap)](q) = sum over the padded positions Q that fold onto pixel q of dy[o](Q - tap)      (27 or 9 rows per head)
//     dmid[c](q)    = sum_rows W3[o][c][tap] g[(o,tap)](q)            then  dz = dmid * act'(mid)   -- head3x3_bwd_data_kernel
//     dW3[o][c][tap]= sum_q g[(o,tap)](q) mid[c](q),   db3[o] = sum_q dy[o](q)                      -- head3x3_bwd_weight_kernel
//
// (the adjoint of pad + 3x3 written on the UNPADDED grid: no (H+2)x(W+2) buffer, no fold pass).  Both are fp32 16x16x4 MFMA
// GEMMs with K = 27 -> 28 (data) / K = pixels (weights) whose operands are gathered straight from dy (6-7 planes, cache
// resident) and mid; both are bound by reading mid once (+ writing dz once).
#include <algorithm>
#include <cstring>
#include <cstdint>
#include "wmd_internal.h"
#include "wmd_head_bwd1.h"

namespace wmd {

constexpr int HB_MAX_SLICES = 24;   // (head, 64-channel slice) work items per launch
constexpr int HB_PART = 8 * 256 + 16;   // floats of one block's partial: 2 row tiles x 4 channel tiles x 16x16 + bias sums

struct HeadBwdSlice {
    int row0;       // first dy plane of the head
    int ch0;        // first mid channel of the head
    int nch;        // channels of the head
    int c_begin;    // first channel (inside the head) of this slice
    int c_count;    // <= 64
    const float* w3;
    float* dw3;
    float* db3;
};

struct HeadBwdK {
    const float* dy3;
    const float* mid;
    float* dz;
    float* partial;
    int B, H, W, Ct, n_out, pad_mode, act;
    float slope;
    int n_slices, nblk;
    HeadBwdSlice s[HB_MAX_SLICES];
};

// padded coordinates (-1 .. n) that fold onto source coordinate q, q itself first
__device__ __forceinline__ int fold_preimage(int q, int n, int pad_mode, int* out) {
    int k = 0;
    out[k++] = q;
    if (pad_mode == WMD_PAD_REFLECT) {
        if (q == 1) out[k++] = -1;
        if (q == n - 2) out[k++] = n;
    } else if (pad_mode == WMD_PAD_REPLICATE) {
        if (q == 0) out[k++] = -1;
        if (q == n - 1) out[k++] = n;
    }
    return k;
}

// g[(o,tap)](q) for one dy plane, general form (maps with fewer than four rows or columns): the forward reads
// midpad(p + (ty-1, tx-1)), so padded position Q receives dy(Q - (ty-1, tx-1))
__device__ __forceinline__ float head_g(const float* __restrict__ dyo, int H, int W, const int* ys, int ny, const int* xs, int nx,
                                        int ty, int tx) {
    float s = 0.f;
    for (int a = 0; a < ny; ++a) {
        const int py = ys[a] - (ty - 1);
        if (py < 0 || py >= H) continue;
        for (int b = 0; b < nx; ++b) {
            const int px = xs[b] - (tx - 1);
            if (px >= 0 && px < W) s += dyo[(size_t)py * W + px];
        }
    }
    return s;
}

// Per-pixel part of g: the <= 2 x 2 padded positions that fold onto pixel q = (qy, qx) (q itself + the ring position next to an
// edge; maps with fewer than 4 rows / columns, where BOTH ring rows can fold onto one source row, take head_g) as base indices
// into a dy plane, and per candidate row / column a 3-bit mask of the taps whose source dy(Q - tap) lies inside the image.
struct HeadPix {
    int base[2][2];
    unsigned rm[2], cm[2];
};
// launch-uniform description of the padding: the source coordinates next to which a ring position folds (lo -> -1, hi -> n)
struct HeadFold {
    int ylo, yhi, xlo, xhi;   // -100 when the mode has no ring (zero padding)
};
__device__ __forceinline__ HeadFold head_fold(int H, int W, int pad_mode) {
    HeadFold f;
    const bool refl = pad_mode == WMD_PAD_REFLECT, none = pad_mode == WMD_PAD_ZERO;
    f.ylo = none ? -100 : (refl ? 1 : 0);
    f.yhi = none ? -100 : (refl ? H - 2 : H - 1);
    f.xlo = none ? -100 : (refl ? 1 : 0);
    f.xhi = none ? -100 : (refl ? W - 2 : W - 1);
    return f;
}
// branch-free (selects only): scalar branches on the padding mode split the gather into ~250 serialised round trips per tile
__device__ __forceinline__ void head_pix(HeadPix& p, int qy, int qx, int H, int W, const HeadFold& f) {
    const bool ey = qy == f.ylo || qy == f.yhi, ex = qx == f.xlo || qx == f.xhi;
    const int y[2] = {qy, ey ? (qy == f.ylo ? -1 : H) : qy}, x[2] = {qx, ex ? (qx == f.xlo ? -1 : W) : qx};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        unsigned r = 0, m = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int py = y[c] - (t - 1), px = x[c] - (t - 1);
            r |= (unsigned)(py >= 0 && py < H) << t;
            m |= (unsigned)(px >= 0 && px < W) << t;
        }
        p.rm[c] = (c == 0 || ey) ? r : 0u;
        p.cm[c] = (c == 0 || ex) ? m : 0u;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) p.base[c][d] = y[c] * W + x[d];
}
// g of N rows (o, ty, tx) at that pixel: all 4 N loads are issued from safe addresses before any value is used (the wave
// issues in order: a load whose result is consumed before the next load is issued costs one memory round trip EACH -- the first
// version of this gather serialised ~250 of them per wave tile and ran at 60 us per tile), selected afterwards; no branches.
template <int N>
__device__ __forceinline__ void head_g_edge(float (&out)[N], const float* __restrict__ dyb, const int (&go)[N], const int (&gty)[N],
                                            const int (&gtx)[N], const bool (&want)[N], const HeadPix& p, int W, int HW) {
    float x[N][4];
    bool ok[N][4];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int toff = go[n] * HW - (gty[n] - 1) * W - (gtx[n] - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                ok[n][2 * c + d] = want[n] && ((p.rm[c] >> gty[n]) & (p.cm[d] >> gtx[n]) & 1u) != 0;
                x[n][2 * c + d] = dyb[ok[n][2 * c + d] ? p.base[c][d] + toff : 0];
            }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) v += ok[n][q] ? x[n][q] : 0.f;
        out[n] = v;
    }
}

// ---- data gradient -----------------------------------------------------------------------------------------------------
// A wave owns 64 consecutive pixels of one frame and walks the <= 4 channel tiles of its slice:
//   D[pixel 16][channel 16] += A[pixel][row 4s+kq] * B[row 4s+kq][channel],   A = g (per pixel group), B = W3 (per channel tile)
// lane (j = l & 15, kq = l >> 4): A operand = g of pixel 16 grp + j for the rows == kq (mod 4), gathered in that layout directly;
// the lane ends up with FOUR CONSECUTIVE PIXELS (4 kq + i) of channel j: 16-byte loads of mid and stores of dz.
template <int NROWS>
__global__ __launch_bounds__(256) void head3x3_bwd_data_kernel(const HeadBwdK a) {
    constexpr int KR = NROWS * 9, KS = (KR + 3) / 4;
    const HeadBwdSlice sl = a.s[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int HW = a.H * a.W, T = (HW + 63) / 64;
    const int nct = (sl.c_count + 15) / 16;
    const bool vec = (HW & 3) == 0;
    const bool tiny = a.H < 4 || a.W < 4;
    const HeadFold fold = head_fold(a.H, a.W, a.pad_mode);
    // act'(mid) in terms of the activation output, branch-free: m > 0 ? 1 : dslope + delu * m  (none: 1, leaky: slope, ELU: 1 + m)
    const float dslope = a.act == WMD_ACT_LEAKY ? a.slope : 1.f, delu = a.act == WMD_ACT_ELU ? 1.f : 0.f;
    // per-lane constants of the rows this lane gathers, and its weight fragments (B operand: channel j of every tile)
    int goff[KS], go[KS], gty[KS], gtx[KS];
    float bw[4][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int r = min(4 * s + kq, KR - 1);
        go[s] = r / 9;
        gty[s] = (r - 9 * go[s]) / 3;
        gtx[s] = (r - 9 * go[s]) % 3;
        goff[s] = go[s] * HW - (gty[s] - 1) * a.W - (gtx[s] - 1);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int c = sl.c_begin + ct * 16 + j;
            bw[ct][s] = (4 * s + kq < KR && ct < nct && c < sl.c_begin + sl.c_count)
                            ? sl.w3[((size_t)go[s] * sl.nch + c) * 9 + gty[s] * 3 + gtx[s]] : 0.f;
        }
    }
    for (int id = blockIdx.x * 4 + wave; id < a.B * T; id += gridDim.x * 4) {
        const int b = id / T, P0 = (id - b * T) * 64;
        const float* dyb = a.dy3 + ((size_t)b * a.n_out + sl.row0) * HW;
        float gA[4][KS];
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int P = P0 + 16 * grp + j;
            const bool valid = P < HW;
            const int qy = valid ? P / a.W : 2, qx = valid ? P - qy * a.W : 2;
            const bool inner = qy >= 2 && qy < a.H - 2 && qx >= 2 && qx < a.W - 2;
            if (__builtin_amdgcn_ballot_w64(valid && !inner) == 0) {
                // every pixel of the group two or more away from every edge: all nine taps in range, nothing folds: ONE load each
#pragma unroll
                for (int s = 0; s < KS; ++s) gA[grp][s] = (valid && 4 * s + kq < KR) ? dyb[goff[s] + P] : 0.f;
            } else if (!tiny) {
                HeadPix pp;
                head_pix(pp, qy, qx, a.H, a.W, fold);
                bool want[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) want[s] = valid && 4 * s + kq < KR;
                head_g_edge<KS>(gA[grp], dyb, go, gty, gtx, want, pp, a.W, HW);
            } else {
                int ys[3], xs[3];
                const int ny = fold_preimage(qy, a.H, a.pad_mode, ys), nx = fold_preimage(qx, a.W, a.pad_mode, xs);
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    gA[grp][s] = (valid && 4 * s + kq < KR) ? head_g(dyb + (size_t)go[s] * HW, a.H, a.W, ys, ny, xs, nx, gty[s], gtx[s]) : 0.f;
            }
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct >= nct) break;
            const int c = sl.c_begin + ct * 16 + j;
            const bool c_ok = c < sl.c_begin + sl.c_count;
            const size_t plane = ((size_t)b * a.Ct + sl.ch0 + min(c, sl.c_begin + sl.c_count - 1)) * HW;
            // the four 16-byte loads of mid are issued before the MFMAs that produce their multiplicands
            float4 mv[4];
            if (vec) {
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    const int P = P0 + 16 * grp + 4 * kq;
                    mv[grp] = (c_ok && P < HW) ? *reinterpret_cast<const float4*>(a.mid + plane + P) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gA[grp][s], bw[ct][s], acc, 0, 0, 0);
                const int P = P0 + 16 * grp + 4 * kq;     // this lane's four pixels of channel c
                if (!c_ok || P >= HW) continue;
                if (vec) {
                    const float4 m = mv[grp];
                    float4 o;
                    o.x = acc[0] * (m.x > 0.f ? 1.f : fmaf(delu, m.x, dslope));
                    o.y = acc[1] * (m.y > 0.f ? 1.f : fmaf(delu, m.y, dslope));
                    o.z = acc[2] * (m.z > 0.f ? 1.f : fmaf(delu, m.z, dslope));
                    o.w = acc[3] * (m.w > 0.f ? 1.f : fmaf(delu, m.w, dslope));
                    *reinterpret_cast<float4*>(a.dz + plane + P) = o;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (P + i < HW) {
                            const float m = a.mid[plane + P + i];
                            a.dz[plane + P + i] = acc[i] * (m > 0.f ? 1.f : fmaf(delu, m, dslope));
                        }
                }
            }
        }
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------
// D[row 16][channel 16] += A[row][pixel] * B[pixel][channel] over the pixels (K index kq of MFMA step 4 g + i = pixel
// 16 g + 4 kq + i of the wave's 64): lane (r16 = l & 15, kq) gathers g of ITS row for four pixels of each 16-pixel group, lane
// (j, kq) loads mid of channel j for the same pixels (four 16-byte loads when the plane allows).  Accumulators stay in registers
// over all tiles of the block.
constexpr int HB_WW = 4;   // wavefronts per block of the weight kernel (256 threads: it also runs as a component of the merged launch)
template <int NROWS>
__device__ __forceinline__ void head3x3_bwd_weight_body(const HeadBwdK& a, int bx, int by, int nbx, float* smem) {
    constexpr int KR = NROWS * 9, RT = (KR + 15) / 16;
    const HeadBwdSlice sl = a.s[by];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int HW = a.H * a.W, T = (HW + 63) / 64;
    const int nct = (sl.c_count + 15) / 16;
    const bool vec = (HW & 3) == 0;
    f32x4 acc[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbs[NROWS];
#pragma unroll
    for (int o = 0; o < NROWS; ++o) dbs[o] = 0.f;
    const bool want_db = sl.c_begin == 0;   // the head's first slice also sums dy
    const bool tiny = a.H < 4 || a.W < 4;
    const HeadFold fold = head_fold(a.H, a.W, a.pad_mode);
    int goff[RT], go[RT], gty[RT], gtx[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int r = min(rt * 16 + j, KR - 1);
        go[rt] = r / 9;
        gty[rt] = (r - 9 * go[rt]) / 3;
        gtx[rt] = (r - 9 * go[rt]) % 3;
        goff[rt] = go[rt] * HW - (gty[rt] - 1) * a.W - (gtx[rt] - 1);
    }

    for (int id = bx * HB_WW + wave; id < a.B * T; id += nbx * HB_WW) {
        const int b = id / T, P0 = (id - b * T) * 64;
        const float* dyb = a.dy3 + ((size_t)b * a.n_out + sl.row0) * HW;
        // K index kq of MFMA step 4 g + i = pixel 16 g + 4 kq + i of the wave's 64 (round 6; before: 16 kq + s): the edge test is
        // then per 16-PIXEL GROUP -- with sixteen consecutive pixels per lane one edge pixel anywhere sent the whole tile down the
        // element-wise path: two tiles of five at W = 320, four of five at W = 160, every tile at W = 80
        float gA[RT][16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int Pj = P0 + 16 * g + j;             // the group's pixels, one per j (the four kq replicas agree)
            const int qyj = Pj / a.W, qxj = Pj - qyj * a.W;
            const bool inner = Pj < HW && qyj >= 2 && qyj < a.H - 2 && qxj >= 2 && qxj < a.W - 2;
            const int Pq = P0 + 16 * g + 4 * kq;        // this lane's four pixels of the group
            if (__builtin_amdgcn_ballot_w64(!inner) == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) gA[rt][4 * g + i] = rt * 16 + j < KR ? dyb[goff[rt] + Pq + i] : 0.f;
            } else {
                int qy = Pq / a.W, qx = Pq - qy * a.W;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool valid = Pq + i < HW;
                    if (!tiny) {
                        HeadPix pp;
                        head_pix(pp, valid ? qy : 2, valid ? qx : 2, a.H, a.W, fold);
                        bool want[RT];
                        float gv[RT];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) want[rt] = valid && rt * 16 + j < KR;
                        head_g_edge<RT>(gv, dyb, go, gty, gtx, want, pp, a.W, HW);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) gA[rt][4 * g + i] = gv[rt];
                    } else {
                        int ys[3], xs[3];
                        const int ny = fold_preimage(qy, a.H, a.pad_mode, ys), nx = fold_preimage(qx, a.W, a.pad_mode, xs);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            gA[rt][4 * g + i] = (valid && rt * 16 + j < KR)
                                                    ? head_g(dyb + (size_t)go[rt] * HW, a.H, a.W, ys, ny, xs, nx, gty[rt], gtx[rt]) : 0.f;
                    }
                    if (++qx == a.W) qx = 0, ++qy;
                }
            }
        }
        if (want_db && P0 + lane < HW) {     // one pixel per lane; the lanes are summed after the loop
#pragma unroll
            for (int o = 0; o < NROWS; ++o) dbs[o] += dyb[(size_t)o * HW + P0 + lane];
        }
        // mid of every channel tile of the slice first (4 x 16-byte loads each, all in flight together), then the MFMAs
        float mB[4][16];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int c = sl.c_begin + ct * 16 + j;
            const bool c_ok = ct < nct && c < sl.c_begin + sl.c_count;
            const float* mp = a.mid + ((size_t)b * a.Ct + sl.ch0 + min(c, sl.c_begin + sl.c_count - 1)) * HW + P0 + 4 * kq;
            if (vec && P0 + 64 <= HW) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = c_ok ? *reinterpret_cast<const float4*>(mp + 16 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
                    mB[ct][4 * g] = v.x, mB[ct][4 * g + 1] = v.y, mB[ct][4 * g + 2] = v.z, mB[ct][4 * g + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mB[ct][4 * g + i] = (c_ok && P0 + 16 * g + 4 * kq + i < HW) ? mp[16 * g + i] : 0.f;
            }
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct >= nct) break;      // wave-uniform; ct itself is a compile-time index of the accumulators
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(gA[rt][s], mB[ct][s], acc[rt][ct], 0, 0, 0);
        }
    }
    // the four waves' accumulators -> one block partial (fixed order), written as [rt][ct][lane][4] + bias sums
    f32x4 (*red)[RT * 4][64] = reinterpret_cast<f32x4 (*)[RT * 4][64]>(smem);                      // [HB_WW - 1][RT * 4][64]
    float (*dbr)[NROWS] = reinterpret_cast<float (*)[NROWS]>(smem + (HB_WW - 1) * RT * 4 * 64 * 4);    // [HB_WW][NROWS]
#pragma unroll
    for (int o = 0; o < NROWS; ++o) {
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) dbs[o] += __shfl_xor(dbs[o], sh);
    }
    if (wave > 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) red[wave - 1][rt * 4 + ct][lane] = acc[rt][ct];
    }
    if (lane == 0) {
#pragma unroll
        for (int o = 0; o < NROWS; ++o) dbr[wave][o] = dbs[o];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.partial + ((size_t)bx * a.n_slices + by) * HB_PART;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                f32x4 v = acc[rt][ct];
                for (int w = 0; w < HB_WW - 1; ++w) v += red[w][rt * 4 + ct][lane];
                *reinterpret_cast<f32x4*>(out + ((rt * 4 + ct) * 64 + lane) * 4) = v;
            }
        if (lane < NROWS) {
            float t = dbr[0][lane];
            for (int w = 1; w < HB_WW; ++w) t += dbr[w][lane];
            out[8 * 256 + lane] = t;
        }
    }
}

// sums the block partials (four threads per element, each a contiguous quarter of the blocks in order, combined in a fixed
// order) and scatters rows (o,tap) x channels into dw3 [nrows, nch, 3, 3] (+ db3).  grid (slices, element chunks of 64)
template <int NROWS>
__device__ __forceinline__ void head3x3_bwd_reduce_body(const HeadBwdK& a, int bx, int by) {
    constexpr int KR = NROWS * 9;
    const HeadBwdSlice sl = a.s[bx];
    const int ne = KR * sl.c_count + (sl.c_begin == 0 ? NROWS : 0);
    const int e = by * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;
    const bool live = e < ne;
    const bool is_db = e >= KR * sl.c_count;
    int src = 0, r = 0, cl = 0;
    if (live) {
        if (is_db) {
            src = 8 * 256 + (e - KR * sl.c_count);
        } else {
            r = e / sl.c_count;
            cl = e - r * sl.c_count;
            // D layout of the 16x16 tile (rt, ct): lane = (col = cl & 15) + 16 * ((r & 15) >> 2), register (r & 15) & 3
            const int rt = r >> 4, ct = cl >> 4, rr = r & 15;
            src = ((rt * 4 + ct) * 64 + (cl & 15) + 16 * (rr >> 2)) * 4 + (rr & 3);
        }
    }
    const int per = (a.nblk + 3) / 4, b0 = q * per, b1 = min(b0 + per, a.nblk);
    float s = 0.f;
    if (live) {
        int blk = b0;
        for (; blk + 4 <= b1; blk += 4) {
            const float p0 = a.partial[((size_t)blk * a.n_slices + bx) * HB_PART + src];
            const float p1 = a.partial[((size_t)(blk + 1) * a.n_slices + bx) * HB_PART + src];
            const float p2 = a.partial[((size_t)(blk + 2) * a.n_slices + bx) * HB_PART + src];
            const float p3 = a.partial[((size_t)(blk + 3) * a.n_slices + bx) * HB_PART + src];
            s = (((s + p0) + p1) + p2) + p3;
        }
        for (; blk < b1; ++blk) s += a.partial[((size_t)blk * a.n_slices + bx) * HB_PART + src];
    }
    const float s1 = __shfl_xor(s, 1);
    const float t = q & 1 ? s1 + s : s + s1;            // both lanes of a pair hold (even + odd) in that order
    const float t2 = __shfl_xor(t, 2);
    const float tot = q & 2 ? t2 + t : t + t2;
    if (live && q == 0) {
        if (is_db) {
            sl.db3[e - KR * sl.c_count] = tot;
        } else {
            const int o = r / 9, tap = r - 9 * o;
            sl.dw3[((size_t)o * sl.nch + sl.c_begin + cl) * 9 + tap] = tot;
        }
    }
}

constexpr int HB_SMEM_FLOATS = (HB_WW - 1) * 8 * 64 * 4 + HB_WW * 4;
template <int NROWS>
__global__ __launch_bounds__(HB_WW * 64) void head3x3_bwd_weight_kernel(const HeadBwdK a) {
    __shared__ __attribute__((aligned(16))) float smem[HB_SMEM_FLOATS];
    head3x3_bwd_weight_body<NROWS>(a, blockIdx.x, blockIdx.y, gridDim.x, smem);
}
template <int NROWS>
__global__ __launch_bounds__(256) void head3x3_bwd_reduce_kernel(const HeadBwdK a) { head3x3_bwd_reduce_body<NROWS>(a, blockIdx.x, blockIdx.y); }

// ---- second stage of wmd_head_bwd as ONE launch ------------------------------------------------------------------------------
// After head3x3_bwd_data_kernel has produced dz, the 3x3 weight gradient, the 1x1 data gradient and the 1x1 weight gradient are
// independent of each other, and each of them alone is a latency chain that leaves most of the GPU idle (20-40 us per launch
// whatever the level's size).  blockIdx.y selects the component (3x3 slices, then the 1x1 data gradient's ci groups, then the 1x1
// weight gradient's channel-group pairs); blockIdx.x beyond a component's own block count returns.  Same for the two reduces.
struct HeadBwdStage2K {
    HeadBwdK h3;
    Head1x1K h1;
    int n3, n1d, n1w;        // blockIdx.y extents of the components
    int nb3, nb1d, nb1w;     // blockIdx.x extents
};
__global__ __launch_bounds__(256) void head_bwd_stage2_kernel(const HeadBwdStage2K m) {
    __shared__ __attribute__((aligned(16))) float smem[H1_SMEM_FLOATS > HB_SMEM_FLOATS ? H1_SMEM_FLOATS : HB_SMEM_FLOATS];
    int y = blockIdx.y;
    if (y < m.n3) {
        if ((int)blockIdx.x < m.nb3) head3x3_bwd_weight_body<3>(m.h3, blockIdx.x, y, m.nb3, smem);
        return;
    }
    y -= m.n3;
    if (y < m.n1d) {
        if ((int)blockIdx.x < m.nb1d) head1x1_bwd_data_body(m.h1, blockIdx.x, y, m.nb1d);
        return;
    }
    y -= m.n1d;
    if ((int)blockIdx.x < m.nb1w) head1x1_bwd_weight_body(m.h1, blockIdx.x, y, m.nb1w, m.n1w, smem);
}
__global__ __launch_bounds__(256) void head_bwd_reduce2_kernel(const HeadBwdStage2K m) {
    if ((int)blockIdx.x < m.n3) {
        if (blockIdx.y < (9 * 3 * 64 + 3 + 63) / 64) head3x3_bwd_reduce_body<3>(m.h3, blockIdx.x, blockIdx.y);
        return;
    }
    head1x1_bwd_reduce_body(m.h1, blockIdx.x - m.n3, blockIdx.y, m.n1w);
}

// ---- both stages of a C = 32 level as ONE pass (round 6) ----------------------------------------------------------------------
// The launch set above moves mid twice, dz three times and x twice (693 MB at the finest level of BASELINE config 2 for 188 MB of
// operands) because its four GEMMs live in four bodies.  At C = 32 (two 3-channel heads of 32 channels: Ct = 64) a wave can hold
// ALL of a 64-pixel tile's work: it gathers g, loads mid ONCE and uses the registers twice (B operand of the 3x3 weight gradient
// with the K index ordered the way the data gradient's accumulators come out: pixel 16 g + 4 kq + i), keeps dz in a wave-private
// LDS tile [64 channels][64 pixels] (row stride 68 floats: every 16-byte access pattern below is conflict-free; dz never reaches
// memory), reads it back as the A operand of the 1x1 data gradient (four consecutive pixels of a channel) and of the 1x1 weight
// gradient (sixteen), and loads x ONCE (the gate of dx, then the B operand of the 1x1 weight gradient).  The 3x3 stage walks
// (head, 16-pixel group) steps on two operand sets: the next step's gathers and mid are in flight behind this step's 30 MFMAs (a
// register rotation by copies would wait for the very loads it copies).  Edge handling is per 16-pixel group (the weight body above
// sends the whole 64-pixel tile down the element-wise path when any lane sees an edge: two tiles of every five at W = 320): the
// data layout is gathered element-wise, the weight layout is its transpose by 28 lane permutes.  No block barrier in the loop.
// Block partials in the layouts of the bodies above: head_bwd_reduce2_kernel finishes them.
// Measured (config 2's finest level, 12 x 96 x 320): 0.139 ms against 0.215 ms for head3x3_bwd_data_kernel + this level's share of
// head_bwd_stage2_kernel.  Where the rest goes (switches since removed): no edge groups 0.116; no mid loads 0.094; no loads at all
// 0.084; the 3x3 stage's instruction stream alone 0.048 for 22.7 us of MFMA issue -- 30 MFMAs of K = 4 per ~65 other instructions.
struct HeadBwdFusedK {
    const float* dy3;
    const float* mid;
    const float* x;
    const float* w1;       // [64][32]
    const float* w3[2];    // [3][32][3][3] per head
    float* dx;
    float* part3;          // [nblk][2][HB_PART]
    float* part1;          // [nblk][H1_PART]
    int B, H, W, n_out, pad_mode;
    int row0[2], ch0[2];
    float m_dslope, m_delu;   // act'(mid) = mid > 0 ? 1 : m_dslope + m_delu * mid
    float x_dslope, x_delu;
};
constexpr int HF_RS = 68;
constexpr int HF_WAVE_FLOATS = 64 * HF_RS;

// head_g_edge for the fused kernel: the same values, two rows (eight loads, eight lane masks) at a time behind scheduling barriers --
// inlined eight times into a kernel that already holds 130 registers, the all-at-once form spilled 350 of them
template <int N>
__device__ __forceinline__ void hf_g_edge(float (&out)[N], const float* __restrict__ dyb, int r_first, int r_step, int r_limit,
                                          const HeadPix& p, int W, int HW) {
#pragma unroll
    for (int n0 = 0; n0 < N; n0 += 2) {
        float x[2][4];
        bool ok[2][4];
#pragma unroll
        for (int n = n0; n < N && n < n0 + 2; ++n) {
            const int r = min(r_first + n * r_step, 26), o = r / 9, ty = (r - 9 * o) / 3, tx = (r - 9 * o) % 3;
            const int toff = o * HW - (ty - 1) * W - (tx - 1);
            const bool want = r_first + n * r_step < r_limit;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    ok[n - n0][2 * c + d] = want && ((p.rm[c] >> ty) & (p.cm[d] >> tx) & 1u) != 0;
                    x[n - n0][2 * c + d] = dyb[ok[n - n0][2 * c + d] ? p.base[c][d] + toff : 0];
                }
        }
#pragma unroll
        for (int n = n0; n < N && n < n0 + 2; ++n) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) v += ok[n - n0][q] ? x[n - n0][q] : 0.f;
            out[n] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// operands of one (head, 16-pixel group) step of the 3x3 stage, requested one step ahead
struct HfPref {
    float gA[7];      // data layout: rows 4 s + kq at pixel 16 g + j
    float gW[2][4];   // weight layout: rows 16 rt + j at pixels 16 g + 4 kq + i
};
struct HfMid {
    float4 v[2];      // mid: channels 16 ct + j of the head at pixels 16 g + 4 kq + i
};

__global__ __launch_bounds__(256, 2) void head_bwd_fused32_kernel(const HeadBwdFusedK a) {
    __shared__ __attribute__((aligned(16))) float smem[4 * HF_WAVE_FLOATS + 32 * HF_RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int H = a.H, W = a.W, HW = H * W, T = HW / 64;
    float* dzs = smem + wave * HF_WAVE_FLOATS;
    float* w1t = smem + 4 * HF_WAVE_FLOATS;      // W1 as the B operand of the 1x1 data gradient: [ci][kq][s], c = 4 s + kq
    const HeadFold fold = head_fold(H, W, a.pad_mode);
    constexpr int KS = 7;     // K-steps of the 27 -> 28 rows of a head
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
        const int c = e >> 5, ci = e & 31;
        w1t[ci * HF_RS + (c & 3) * 16 + (c >> 2)] = a.w1[e];
    }

    // rows this lane gathers: data layout (A[pixel][row 4 s + kq]) and weight layout (A[row 16 rt + j][pixel])
    int goff[KS], goffW[2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int r = min(4 * s + kq, 26), o = r / 9, ty = (r - 9 * o) / 3, tx = (r - 9 * o) % 3;
        goff[s] = o * HW - (ty - 1) * W - (tx - 1);
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = min(rt * 16 + j, 26), o = r / 9, ty = (r - 9 * o) / 3, tx = (r - 9 * o) % 3;
        goffW[rt] = o * HW - (ty - 1) * W - (tx - 1);
    }
    // W3 as the B operand of the data gradient: row 4 s + kq, channel 16 ct + j of head h
    float bw[2][2][KS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int r = min(4 * s + kq, 26), o = r / 9, tap = r - 9 * o;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) bw[h][ct][s] = 4 * s + kq < 27 ? a.w3[h][((size_t)o * 32 + ct * 16 + j) * 9 + tap] : 0.f;
        }

    f32x4 acc3[2][2][2], acc1[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc3[h][rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc1[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db3s[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, db1s[4] = {0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy3), 0, a.B * a.n_out * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmid = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mid), 0, a.B * 64 * HW * 4, 0x00020000);
    const unsigned mid_lane = (unsigned)(j * HW + 4 * kq) * 4u;
    // requests the operands of step (head hh, group g) of tile (b, P0); edge handling per 16-pixel group (wave-uniform: the four kq
    // replicas of a pixel agree).  The element-wise path consumes its loads on the spot (a stall on ~12 % of the groups).
    auto fetch = [&](HfPref& p, int b, int P0, int hh, int g) __attribute__((always_inline)) {
        const float* dyb = a.dy3 + ((size_t)b * a.n_out + (hh ? a.row0[1] : a.row0[0])) * HW;
        const int P = P0 + 16 * g + j, qy = P / W, qx = P - qy * W;
        const bool inner = qy >= 2 && qy < H - 2 && qx >= 2 && qx < W - 2;
        if (__builtin_amdgcn_ballot_w64(!inner) == 0) {
            // buffer loads: the plane's base is a scalar offset, the lane adds ONE 32-bit offset per load.  Rows past the 27th (the
            // pad of the K index / of the last row tile) read row 26 again: their products meet a zero weight (data gradient) or
            // land in rows of the weight tile that the reduce never reads -- no selects.
            const unsigned soff = (unsigned)((b * a.n_out + (hh ? a.row0[1] : a.row0[0])) * HW) * 4u;
            const int P4 = P * 4;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                p.gA[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, (unsigned)(goff[s] * 4 + P4), soff, 0));
            const int Pw4 = (P0 + 16 * g + 4 * kq) * 4;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, (unsigned)(goffW[rt] * 4 + Pw4), soff, 0));
                p.gW[rt][0] = v[0], p.gW[rt][1] = v[1], p.gW[rt][2] = v[2], p.gW[rt][3] = v[3];
            }
        } else {
            {
                HeadPix pp;
                head_pix(pp, qy, qx, H, W, fold);
                hf_g_edge<KS>(p.gA, dyb, kq, 4, 27, pp, W, HW);          // rows 4 s + kq
            }
            // weight layout = the transpose of what the wave now holds: g[row 16 rt + j][pixel 4 kq + i] is register (16 rt + j) >> 2
            // of lane (4 kq + i) + 16 (j & 3) -- 28 lane permutes instead of four more element-wise gathers
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int src = 4 * kq + i + 16 * (j & 3);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    float v = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (4 * rt + q < KS) {
                            const float t = __shfl(p.gA[4 * rt + q], src);
                            v = (j >> 2) == q ? t : v;
                        }
                    }
                    p.gW[rt][i] = rt * 16 + j < 27 ? v : 0.f;
                }
            }
        }
    };

    auto fetch_mid = [&](HfMid& m, int b, int P0, int hh, int g) __attribute__((always_inline)) {
        // scalar offset: the head's first plane + the group; lane offset: its channel's plane + its four pixels (a constant)
        const unsigned soff = (unsigned)((b * 64 + (hh ? a.ch0[1] : a.ch0[0])) * HW + P0 + 16 * g) * 4u;
        const f32x4 v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rmid, mid_lane, soff, 0));
        const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rmid, mid_lane, soff + (unsigned)(16 * HW) * 4u, 0));
        m.v[0] = make_float4(v0[0], v0[1], v0[2], v0[3]);
        m.v[1] = make_float4(v1[0], v1[1], v1[2], v1[3]);
    };

    __syncthreads();     // w1t is complete
    const int ntile = a.B * T, stride = gridDim.x * 4;
    int id = blockIdx.x * 4 + wave;
    HfPref pa, pb;       // operand sets of the even / odd steps
    HfMid ma, mb;
    if (id < ntile) {
        const int b = id / T, P0 = (id - b * T) * 64;
        fetch_mid(ma, b, P0, 0, 0);
        fetch(pa, b, P0, 0, 0);
    }
    for (; id < ntile; id += stride) {
        const int b = id / T, P0 = (id - b * T) * 64;
        const int idn = id + stride, bn = idn / T, P0n = (idn - bn * T) * 64;
        float4 xv[2][4];
        // ---- 3x3 stage: one head, one 16-pixel group per step; the next step's operands are in flight behind this one's MFMAs ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float* dyb = a.dy3 + ((size_t)b * a.n_out + a.row0[h]) * HW;
            float dbv[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) dbv[o] = dyb[(size_t)o * HW + P0 + lane];
            // two steps per iteration on two operand sets (A, B): a register rotation by copies would wait for the loads it copies
            auto step = [&](const HfPref& c, const HfMid& cm, HfPref& n, HfMid& nm, int g) __attribute__((always_inline)) {
                {   // the successor of step (h, g): the next group, the other head, the next tile
                    int nb = b, nP0 = P0, nh = h, ng = g + 1;
                    if (g == 3) {
                        ng = 0;
                        if (h == 0) nh = 1;
                        else nb = bn, nP0 = P0n, nh = 0;
                    }
                    if (h == 0 || g < 3) {      // (the next tile's first step is requested behind the 1x1 data gradient, below)
                        fetch_mid(nm, nb, nP0, nh, ng);
                        fetch(n, nb, nP0, nh, ng);
                    } else {                    // the tile's last step requests x: the gate of dx, then the B operand of the 1x1 weight gradient
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const float* xp = a.x + ((size_t)b * 32 + ct * 16 + j) * HW + P0 + 16 * kq;
#pragma unroll
                            for (int i = 0; i < 4; ++i) xv[ct][i] = *reinterpret_cast<const float4*>(xp + 4 * i);
                        }
                    }
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const float4 m = cm.v[ct];
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.gA[s], bw[h][ct][s], acc, 0, 0, 0);
                    f32x4 o;
                    o[0] = acc[0] * (m.x > 0.f ? 1.f : fmaf(a.m_delu, m.x, a.m_dslope));
                    o[1] = acc[1] * (m.y > 0.f ? 1.f : fmaf(a.m_delu, m.y, a.m_dslope));
                    o[2] = acc[2] * (m.z > 0.f ? 1.f : fmaf(a.m_delu, m.z, a.m_dslope));
                    o[3] = acc[3] * (m.w > 0.f ? 1.f : fmaf(a.m_delu, m.w, a.m_dslope));
                    *reinterpret_cast<f32x4*>(dzs + (a.ch0[h] + ct * 16 + j) * HF_RS + 16 * g + 4 * kq) = o;
                    // 3x3 weight gradient: K index kq of step (g, i) = pixel 16 g + 4 kq + i -- what m holds
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        acc3[h][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.gW[rt][0], m.x, acc3[h][rt][ct], 0, 0, 0);
                        acc3[h][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.gW[rt][1], m.y, acc3[h][rt][ct], 0, 0, 0);
                        acc3[h][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.gW[rt][2], m.z, acc3[h][rt][ct], 0, 0, 0);
                        acc3[h][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.gW[rt][3], m.w, acc3[h][rt][ct], 0, 0, 0);
                    }
                }
            };
#pragma unroll 1
            for (int gp = 0; gp < 2; ++gp) {
                step(pa, ma, pb, mb, 2 * gp);
                step(pb, mb, pa, ma, 2 * gp + 1);
            }
#pragma unroll
            for (int o = 0; o < 3; ++o) db3s[h][o] += dbv[o];
        }
        // ---- 1x1 data gradient: D[pixel 4 j' + g][ci] += dz[pixel][c = 4 s + kq] * W1[c][ci]; one 16-channel tile of ci at a time
        //      (both at once: 32 accumulators more than the register file has left) ----
        {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f32x4 accB[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) accB[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sq = 0; sq < 4; ++sq) {
                    const f32x4 wq = *reinterpret_cast<const f32x4*>(w1t + (ct * 16 + j) * HF_RS + kq * 16 + 4 * sq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int s = 4 * sq + e;
                        const f32x4 A = *reinterpret_cast<const f32x4*>(dzs + (4 * s + kq) * HF_RS + 4 * j);
#pragma unroll
                        for (int g = 0; g < 4; ++g) accB[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g], wq[e], accB[g], 0, 0, 0);
                    }
                }
                // register i of group g = pixel 4 (4 kq + i) + g: a float4 over g is four consecutive pixels of channel 16 ct + j
                float* dp = a.dx + ((size_t)b * 32 + ct * 16 + j) * HW + P0 + 16 * kq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 m = xv[ct][i];
                    float4 o;
                    o.x = accB[0][i] * (m.x > 0.f ? 1.f : fmaf(a.x_delu, m.x, a.x_dslope));
                    o.y = accB[1][i] * (m.y > 0.f ? 1.f : fmaf(a.x_delu, m.y, a.x_dslope));
                    o.z = accB[2][i] * (m.z > 0.f ? 1.f : fmaf(a.x_delu, m.z, a.x_dslope));
                    o.w = accB[3][i] * (m.w > 0.f ? 1.f : fmaf(a.x_delu, m.w, a.x_dslope));
                    *reinterpret_cast<float4*>(dp + 4 * i) = o;
                }
            }
        }
        if (idn < ntile) {       // the next tile's first step: in flight behind the 128 MFMAs below
            fetch_mid(ma, bn, P0n, 0, 0);
            fetch(pa, bn, P0n, 0, 0);
        }
        // ---- 1x1 weight gradient: D[c][ci] += dz[c = 16 rt + j][pixel 16 kq + s] * x[pixel][ci] ----
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            f32x4 A[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) A[q] = *reinterpret_cast<const f32x4*>(dzs + (rt * 16 + j) * HF_RS + 16 * kq + 4 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) db1s[rt] += (A[q][0] + A[q][1]) + (A[q][2] + A[q][3]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc1[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q][0], xv[ct][q].x, acc1[rt][ct], 0, 0, 0);
                    acc1[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q][1], xv[ct][q].y, acc1[rt][ct], 0, 0, 0);
                    acc1[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q][2], xv[ct][q].z, acc1[rt][ct], 0, 0, 0);
                    acc1[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q][3], xv[ct][q].w, acc1[rt][ct], 0, 0, 0);
                }
        }
    }

    // ---- block partials (the layouts of head3x3_bwd_weight_body / head1x1_bwd_weight_body) ----
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int o = 0; o < 3; ++o) {
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) db3s[h][o] += __shfl_xor(db3s[h][o], sh);
        }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        db1s[rt] += __shfl_xor(db1s[rt], 16);
        db1s[rt] += __shfl_xor(db1s[rt], 32);
    }
    __syncthreads();     // every wave is done with its dz tile: the space becomes the reduction buffer
    f32x4 (*red)[16][64] = reinterpret_cast<f32x4 (*)[16][64]>(smem);            // [3][16 tiles][64 lanes]
    float (*dbr3)[8] = reinterpret_cast<float (*)[8]>(smem + 3 * 16 * 64 * 4);    // [4][8]
    float (*dbr1)[64] = reinterpret_cast<float (*)[64]>(smem + 3 * 16 * 64 * 4 + 32);   // [4][64]
    if (wave > 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) red[wave - 1][h * 4 + rt * 2 + ct][lane] = acc3[h][rt][ct];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) red[wave - 1][8 + rt * 2 + ct][lane] = acc1[rt][ct];
    }
    if (lane == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int o = 0; o < 3; ++o) dbr3[wave][h * 3 + o] = db3s[h][o];
    }
    if (kq == 0) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) dbr1[wave][rt * 16 + j] = db1s[rt];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float* out = a.part3 + ((size_t)blockIdx.x * 2 + h) * HB_PART;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    f32x4 v = acc3[h][rt][ct];
                    for (int w = 0; w < 3; ++w) v += red[w][h * 4 + rt * 2 + ct][lane];
                    *reinterpret_cast<f32x4*>(out + ((rt * 4 + ct) * 64 + lane) * 4) = v;
                }
            if (lane < 3) {
                float t = dbr3[0][h * 3 + lane];
                for (int w = 1; w < 4; ++w) t += dbr3[w][h * 3 + lane];
                out[8 * 256 + lane] = t;
            }
        }
        float* out = a.part1 + (size_t)blockIdx.x * H1_PART;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f32x4 v = acc1[rt][ct];
                for (int w = 0; w < 3; ++w) v += red[w][8 + rt * 2 + ct][lane];
                *reinterpret_cast<f32x4*>(out + ((rt * 4 + ct) * 64 + lane) * 4) = v;
            }
        float t = dbr1[0][lane];
        for (int w = 1; w < 4; ++w) t += dbr1[w][lane];
        out[16 * 256 + lane] = t;
    }
}

int head1x1_blocks(const wmd_head1x1_bwd_args* g);
int head1x1_validate(const wmd_head1x1_bwd_args* g);
void head1x1_fill(const wmd_head1x1_bwd_args* g, Head1x1K* a);

static int head_bwd_blocks(const wmd_head3x3_bwd_args* g) {
    const long tiles = (long)g->B * (((long)g->H * g->W + 63) / 64);
    return (int)std::max<long>(1, std::min<long>((tiles + HB_WW - 1) / HB_WW, 256));
}
// blocks the workspace is sized for: the fused C = 32 launch runs the 1x1 stage's block count (two blocks per CU)
static int head_bwd_ws_blocks(const wmd_head3x3_bwd_args* g) {
    const long tiles = (long)g->B * (((long)g->H * g->W + 63) / 64);
    return std::max(head_bwd_blocks(g), (int)std::max<long>(1, std::min<long>((tiles + H1_WW - 1) / H1_WW, H1_MAX_BLK)));
}

}  // namespace wmd

using namespace wmd;

static int head_bwd_validate(const wmd_head3x3_bwd_args* g, int* n_slices) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_bwd: null args");
    if (!g->dy3 || !g->mid || !g->dzmid) return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_bwd: null tensor pointer");
    if (g->B <= 0 || g->H <= 0 || g->W <= 0 || g->Ct <= 0 || g->n_out <= 0)
        return fail(WMD_ERR_BAD_SHAPE, "wmd_head3x3_bwd: B=%d H=%d W=%d Ct=%d n_out=%d", g->B, g->H, g->W, g->Ct, g->n_out);
    if (g->pad_mode < 0 || g->pad_mode > 2) return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_bwd: pad_mode=%d", g->pad_mode);
    if (g->act != WMD_ACT_NONE && g->act != WMD_ACT_LEAKY && g->act != WMD_ACT_ELU)
        return fail(WMD_ERR_UNSUPPORTED, "wmd_head3x3_bwd: act=%d (none, LeakyReLU or ELU)", g->act);
    if (g->pad_mode == WMD_PAD_REFLECT && (g->H < 2 || g->W < 2))
        return fail(WMD_ERR_BAD_SHAPE, "wmd_head3x3_bwd: reflection padding needs H, W >= 2 (got %dx%d)", g->H, g->W);
    if (g->n_heads < 1 || g->n_heads > 3) return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_bwd: n_heads=%d", g->n_heads);
    int n = 0;
    for (int k = 0; k < g->n_heads; ++k) {
        const wmd_head_bwd_head& h = g->head[k];
        if (!h.w3 || !h.dw3 || !h.db3) return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_bwd: head %d: null pointer", k);
        if ((h.nrows != 1 && h.nrows != 3) || h.nch <= 0 || h.row0 < 0 || h.row0 + h.nrows > g->n_out || h.ch0 < 0 ||
            h.ch0 + h.nch > g->Ct)
            return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_bwd: head %d: rows %d+%d of %d, channels %d+%d of %d", k, h.row0, h.nrows,
                        g->n_out, h.ch0, h.nch, g->Ct);
        n += (h.nch + 63) / 64;
    }
    if (n > HB_MAX_SLICES) return fail(WMD_ERR_UNSUPPORTED, "wmd_head3x3_bwd: %d channel slices (at most %d)", n, HB_MAX_SLICES);
    *n_slices = n;
    return WMD_OK;
}

extern "C" size_t wmd_head3x3_bwd_workspace_floats(const wmd_head3x3_bwd_args* g) {
    int n = 0;
    if (head_bwd_validate(g, &n)) return 0;
    return (size_t)head_bwd_ws_blocks(g) * n * HB_PART;
}

// fills the kernel arguments of the launch set for heads with `nrows` output channels; returns the floats of partials it uses
static size_t head_bwd_fill(const wmd_head3x3_bwd_args* g, int nrows, int nblk, float* ws, HeadBwdK* a, double* ch_out) {
    memset(a, 0, sizeof(*a));
    a->dy3 = g->dy3;
    a->mid = g->mid;
    a->dz = g->dzmid;
    a->B = g->B, a->H = g->H, a->W = g->W, a->Ct = g->Ct, a->n_out = g->n_out, a->pad_mode = g->pad_mode, a->act = g->act;
    a->slope = g->slope;
    a->nblk = nblk;
    double ch = 0;
    for (int k = 0; k < g->n_heads; ++k) {
        const wmd_head_bwd_head& h = g->head[k];
        if (h.nrows != nrows) continue;
        for (int c0 = 0; c0 < h.nch; c0 += 64) {
            HeadBwdSlice& sl = a->s[a->n_slices++];
            sl.row0 = h.row0, sl.ch0 = h.ch0, sl.nch = h.nch, sl.c_begin = c0, sl.c_count = std::min(64, h.nch - c0);
            sl.w3 = h.w3, sl.dw3 = h.dw3, sl.db3 = h.db3;
        }
        ch += h.nch;
    }
    a->partial = ws;
    *ch_out = ch;
    return (size_t)nblk * a->n_slices * HB_PART;
}

static int head_bwd_launch_data(const wmd_head3x3_bwd_args* g, const HeadBwdK& a, int nrows, double ch, hipStream_t s) {
    const double pix = (double)g->B * g->H * g->W;
    ProfScope prof("head3x3_bwd_data_kernel", 2.0 * 9 * nrows * ch * pix, 4.0 * pix * (2.0 * ch + nrows), s);
    const long tiles = (long)g->B * (((long)g->H * g->W + 63) / 64);
    const dim3 dgrid((unsigned)std::max<long>(1, std::min<long>((tiles + 3) / 4, 4096)), a.n_slices);   // one tile per wave
    if (nrows == 3) hipLaunchKernelGGL(head3x3_bwd_data_kernel<3>, dgrid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(head3x3_bwd_data_kernel<1>, dgrid, dim3(256), 0, s, a);
    return check_launch("head3x3_bwd_data_kernel");
}

extern "C" int wmd_head3x3_bwd(const wmd_head3x3_bwd_args* g, void* stream) {
    int n_all = 0;
    if (int st = head_bwd_validate(g, &n_all)) return st;
    const int nblk = head_bwd_blocks(g);
    if (!g->workspace || g->workspace_floats < (size_t)nblk * n_all * HB_PART)
        return fail(WMD_ERR_WORKSPACE, "wmd_head3x3_bwd: workspace %zu < %zu floats", g->workspace_floats, (size_t)nblk * n_all * HB_PART);
    hipStream_t s = (hipStream_t)stream;
    const double pix = (double)g->B * g->H * g->W;
    float* ws = g->workspace;
    for (int nrows = 3; nrows >= 1; nrows -= 2) {   // one launch set per head kind: 3-row (+/-) heads, then the 1-row low-pass head
        HeadBwdK a;
        double ch = 0;
        ws += head_bwd_fill(g, nrows, nblk, ws, &a, &ch);
        if (a.n_slices == 0) continue;
        const dim3 grid(nblk, a.n_slices);
        if (int st = head_bwd_launch_data(g, a, nrows, ch, s)) return st;
        {
            ProfScope prof("head3x3_bwd_weight_kernel", 2.0 * 9 * nrows * ch * pix, 4.0 * pix * (ch + nrows), s);
            if (nrows == 3) hipLaunchKernelGGL(head3x3_bwd_weight_kernel<3>, grid, dim3(HB_WW * 64), 0, s, a);
            else hipLaunchKernelGGL(head3x3_bwd_weight_kernel<1>, grid, dim3(HB_WW * 64), 0, s, a);
        }
        if (int st = check_launch("head3x3_bwd_weight_kernel")) return st;
        {
            ProfScope prof("head3x3_bwd_reduce_kernel", (double)nblk * a.n_slices * HB_PART, 4.0 * nblk * a.n_slices * HB_PART, s);
            const dim3 rgrid(a.n_slices, (9 * nrows * 64 + nrows + 63) / 64);
            if (nrows == 3) hipLaunchKernelGGL(head3x3_bwd_reduce_kernel<3>, rgrid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL(head3x3_bwd_reduce_kernel<1>, rgrid, dim3(256), 0, s, a);
        }
        if (int st = check_launch("head3x3_bwd_reduce_kernel")) return st;
    }
    return WMD_OK;
}

// may both stages run as head_bwd_fused32_kernel?  (two 3-channel heads of 32 channels each over a 32-channel x, whole 64-pixel
// tiles, rows that are multiples of four pixels; WMD_HEAD_BWD_FUSED=0: never)
static bool head_bwd_fused_ok(const wmd_head3x3_bwd_args* a3, const wmd_head1x1_bwd_args* a1) {
    static const bool on = !(getenv("WMD_HEAD_BWD_FUSED") && atoi(getenv("WMD_HEAD_BWD_FUSED")) == 0);
    if (!on || a3->n_heads != 2 || a3->Ct != 64 || a1->C != 32 || !a1->dx || a3->pad_mode != WMD_PAD_REFLECT) return false;   // (the decoders' padding: what the tests cover)
    for (int k = 0; k < 2; ++k)
        if (a3->head[k].nrows != 3 || a3->head[k].nch != 32 || (a3->head[k].ch0 != 0 && a3->head[k].ch0 != 32)) return false;
    if (a3->head[0].ch0 == a3->head[1].ch0) return false;
    const long HW = (long)a3->H * a3->W;
    // 16-byte accesses of mid, x and dx; 32-bit byte offsets into dy3 and mid (buffer loads)
    if ((((uintptr_t)a3->mid | (uintptr_t)a1->x | (uintptr_t)a1->dx) & 15) != 0) return false;
    if ((double)a3->B * 64 * HW * 4 >= 2147483648.0) return false;
    return a3->H >= 4 && a3->W >= 4 && a3->W % 4 == 0 && HW % 64 == 0;
}

// Both stages of a level's heads in three launches: the 3x3 data gradient (-> dzmid), then the 3x3 weight gradient + the 1x1
// data gradient + the 1x1 weight gradient as ONE launch, then both reduces as one.  a1->dz must be a3->dzmid.
// Round 6, C = 32 levels: ONE launch for all four GEMMs (head_bwd_fused32_kernel; dzmid is then not written) + the reduce.
extern "C" int wmd_head_bwd(const wmd_head3x3_bwd_args* a3, const wmd_head1x1_bwd_args* a1, void* stream) {
    int n_all = 0;
    if (int st = head_bwd_validate(a3, &n_all)) return st;
    if (int st = head1x1_validate(a1)) return st;
    for (int k = 0; k < a3->n_heads; ++k)
        if (a3->head[k].nrows != 3) return fail(WMD_ERR_UNSUPPORTED, "wmd_head_bwd: 3-channel heads only (the low-pass head: wmd_head3x3_bwd + wmd_head1x1_bwd)");
    if (a1->dz != a3->dzmid || a1->B != a3->B || a1->H != a3->H || a1->W != a3->W || a1->Ct != a3->Ct)
        return fail(WMD_ERR_BAD_ARG, "wmd_head_bwd: the 1x1 stage must consume the 3x3 stage's dzmid (same B, H, W, Ct)");
    const bool fused = head_bwd_fused_ok(a3, a1) && a3->workspace && a3->workspace_floats >= (size_t)head1x1_blocks(a1) * n_all * HB_PART;
    const int nblk3 = fused ? head1x1_blocks(a1) : head_bwd_blocks(a3);
    if (!a3->workspace || a3->workspace_floats < (size_t)nblk3 * n_all * HB_PART)
        return fail(WMD_ERR_WORKSPACE, "wmd_head_bwd: 3x3 workspace %zu < %zu floats", a3->workspace_floats, (size_t)nblk3 * n_all * HB_PART);
    HeadBwdStage2K m;
    memset(&m, 0, sizeof(m));
    double ch = 0;
    head_bwd_fill(a3, 3, nblk3, a3->workspace, &m.h3, &ch);
    head1x1_fill(a1, &m.h1);
    const size_t need1 = (size_t)m.h1.nblk * m.h1.n_cgrp * m.h1.n_igrp * H1_PART;
    if (!a1->workspace || a1->workspace_floats < need1)
        return fail(WMD_ERR_WORKSPACE, "wmd_head_bwd: 1x1 workspace %zu < %zu floats", a1->workspace_floats, need1);
    hipStream_t s = (hipStream_t)stream;
    const double pix = (double)a3->B * a3->H * a3->W;
    const long tiles = (long)a3->B * (((long)a3->H * a3->W + 63) / 64);
    m.n3 = m.h3.n_slices;
    m.n1d = a1->dx ? m.h1.n_igrp : 0;
    m.n1w = m.h1.n_cgrp * m.h1.n_igrp;
    m.nb3 = nblk3;
    m.nb1d = (int)std::max<long>(1, std::min<long>((tiles + 3) / 4, 1024));
    m.nb1w = m.h1.nblk;
    if (fused) {
        HeadBwdFusedK f;
        memset(&f, 0, sizeof(f));
        f.dy3 = a3->dy3, f.mid = a3->mid, f.x = a1->x, f.w1 = a1->w1, f.dx = a1->dx;
        f.part3 = m.h3.partial, f.part1 = m.h1.partial;
        f.B = a3->B, f.H = a3->H, f.W = a3->W, f.n_out = a3->n_out, f.pad_mode = a3->pad_mode;
        for (int k = 0; k < 2; ++k) f.w3[k] = a3->head[k].w3, f.row0[k] = a3->head[k].row0, f.ch0[k] = a3->head[k].ch0;
        f.m_dslope = a3->act == WMD_ACT_LEAKY ? a3->slope : 1.f, f.m_delu = a3->act == WMD_ACT_ELU ? 1.f : 0.f;
        f.x_dslope = m.h1.dslope, f.x_delu = m.h1.delu;
        ProfScope prof("head_bwd_fused32_kernel", 2.0 * pix * (2.0 * 27 * 64 + 2.0 * 64 * 32), 4.0 * pix * (64 + 32 + 32 + a3->n_out), s);
        hipLaunchKernelGGL(head_bwd_fused32_kernel, dim3(nblk3), dim3(256), 0, s, f);
        if (int st = check_launch("head_bwd_fused32_kernel")) return st;
    } else {
    if (int st = head_bwd_launch_data(a3, m.h3, 3, ch, s)) return st;
    {
        ProfScope prof("head_bwd_stage2_kernel", 2.0 * pix * (27.0 * ch + 2.0 * a1->C * a1->Ct),
                       4.0 * pix * (ch + 3.0 + 2.0 * a1->Ct + 3.0 * a1->C), s);
        hipLaunchKernelGGL(head_bwd_stage2_kernel, dim3(std::max(m.nb3, std::max(m.nb1d, m.nb1w)), m.n3 + m.n1d + m.n1w), dim3(256), 0, s, m);
    }
    if (int st = check_launch("head_bwd_stage2_kernel")) return st;
    }
    {
        ProfScope prof("head_bwd_reduce2_kernel", (double)nblk3 * m.n3 * HB_PART + (double)need1, 4.0 * (nblk3 * m.n3 * HB_PART + need1), s);
        hipLaunchKernelGGL(head_bwd_reduce2_kernel, dim3(m.n3 + m.n1w, (64 * 64 + 64 + 15) / 16), dim3(256), 0, s, m);
    }
    return check_launch("head_bwd_reduce2_kernel");
}
