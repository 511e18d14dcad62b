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
// D[row 16][channel 16] += A[row][pixel] * B[pixel][channel] over the pixels (K index kq of MFMA step s = pixel 16 kq + s of the
// wave's 64): lane (r16 = l & 15, kq) gathers g of ITS row for 16 consecutive pixels, lane (j, kq) loads mid of channel j for
// the same 16 pixels (four 16-byte loads when the plane allows).  Accumulators stay in registers over all tiles of the block.
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
        const int Pl = P0 + 16 * kq;     // this lane's 16 pixels
        float gA[RT][16];
        {
            int qy = Pl / a.W, qx = Pl - qy * a.W;     // walks the 16 pixels in raster order
            // the lane's 16 pixels are all two or more away from every edge <=> one row, columns 2 .. W-3, rows 2 .. H-3
            const bool inner = Pl + 16 <= HW && qy >= 2 && qy < a.H - 2 && qx >= 2 && qx + 15 < a.W - 2;
            if (__builtin_amdgcn_ballot_w64(!inner) == 0) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) gA[rt][s] = rt * 16 + j < KR ? dyb[goff[rt] + Pl + s] : 0.f;
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const bool valid = Pl + s < HW;
                    if (!tiny) {
                        HeadPix pp;
                        head_pix(pp, valid ? qy : 2, valid ? qx : 2, a.H, a.W, fold);
                        bool want[RT];
                        float gv[RT];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) want[rt] = valid && rt * 16 + j < KR;
                        head_g_edge<RT>(gv, dyb, go, gty, gtx, want, pp, a.W, HW);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) gA[rt][s] = gv[rt];
                    } else {
                        int ys[3], xs[3];
                        const int ny = fold_preimage(qy, a.H, a.pad_mode, ys), nx = fold_preimage(qx, a.W, a.pad_mode, xs);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            gA[rt][s] = (valid && rt * 16 + j < KR)
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
            const float* mp = a.mid + ((size_t)b * a.Ct + sl.ch0 + min(c, sl.c_begin + sl.c_count - 1)) * HW + Pl;
            if (vec && Pl + 16 <= HW) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 v = c_ok ? *reinterpret_cast<const float4*>(mp + 4 * q4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    mB[ct][4 * q4] = v.x, mB[ct][4 * q4 + 1] = v.y, mB[ct][4 * q4 + 2] = v.z, mB[ct][4 * q4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) mB[ct][s] = (c_ok && Pl + s < HW) ? mp[s] : 0.f;
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

int head1x1_blocks(const wmd_head1x1_bwd_args* g);
int head1x1_validate(const wmd_head1x1_bwd_args* g);
void head1x1_fill(const wmd_head1x1_bwd_args* g, Head1x1K* a);

static int head_bwd_blocks(const wmd_head3x3_bwd_args* g) {
    const long tiles = (long)g->B * (((long)g->H * g->W + 63) / 64);
    return (int)std::max<long>(1, std::min<long>((tiles + HB_WW - 1) / HB_WW, 256));
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
    return (size_t)head_bwd_blocks(g) * n * HB_PART;
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

// Both stages of a level's heads in three launches: the 3x3 data gradient (-> dzmid), then the 3x3 weight gradient + the 1x1
// data gradient + the 1x1 weight gradient as ONE launch, then both reduces as one.  a1->dz must be a3->dzmid.
extern "C" int wmd_head_bwd(const wmd_head3x3_bwd_args* a3, const wmd_head1x1_bwd_args* a1, void* stream) {
    int n_all = 0;
    if (int st = head_bwd_validate(a3, &n_all)) return st;
    if (int st = head1x1_validate(a1)) return st;
    for (int k = 0; k < a3->n_heads; ++k)
        if (a3->head[k].nrows != 3) return fail(WMD_ERR_UNSUPPORTED, "wmd_head_bwd: 3-channel heads only (the low-pass head: wmd_head3x3_bwd + wmd_head1x1_bwd)");
    if (a1->dz != a3->dzmid || a1->B != a3->B || a1->H != a3->H || a1->W != a3->W || a1->Ct != a3->Ct)
        return fail(WMD_ERR_BAD_ARG, "wmd_head_bwd: the 1x1 stage must consume the 3x3 stage's dzmid (same B, H, W, Ct)");
    const int nblk3 = head_bwd_blocks(a3);
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
    if (int st = head_bwd_launch_data(a3, m.h3, 3, ch, s)) return st;
    const double pix = (double)a3->B * a3->H * a3->W;
    const long tiles = (long)a3->B * (((long)a3->H * a3->W + 63) / 64);
    m.n3 = m.h3.n_slices;
    m.n1d = a1->dx ? m.h1.n_igrp : 0;
    m.n1w = m.h1.n_cgrp * m.h1.n_igrp;
    m.nb3 = nblk3;
    m.nb1d = (int)std::max<long>(1, std::min<long>((tiles + 3) / 4, 1024));
    m.nb1w = m.h1.nblk;
    {
        ProfScope prof("head_bwd_stage2_kernel", 2.0 * pix * (27.0 * ch + 2.0 * a1->C * a1->Ct),
                       4.0 * pix * (ch + 3.0 + 2.0 * a1->Ct + 3.0 * a1->C), s);
        hipLaunchKernelGGL(head_bwd_stage2_kernel, dim3(std::max(m.nb3, std::max(m.nb1d, m.nb1w)), m.n3 + m.n1d + m.n1w), dim3(256), 0, s, m);
    }
    if (int st = check_launch("head_bwd_stage2_kernel")) return st;
    {
        ProfScope prof("head_bwd_reduce2_kernel", (double)nblk3 * m.n3 * HB_PART + (double)need1, 4.0 * (nblk3 * m.n3 * HB_PART + need1), s);
        hipLaunchKernelGGL(head_bwd_reduce2_kernel, dim3(m.n3 + m.n1w, (64 * 64 + 64 + 15) / 16), dim3(256), 0, s, m);
    }
    return check_launch("head_bwd_reduce2_kernel");
}
