// Device code of the backward of the 1x1 stage of a level's wavelet heads (training path) for gfx950.
//
// Reference: the heads' first layer Conv1x1(C, C | C/4) + LeakyReLU of every head of a level reads the same decoder feature x
// (KITTI/networks/decoders/depth_decoder.py:104-136); stacked along the output channels it is ONE [Ct, C] GEMM per pixel.  Its
// backward (torch.autograd in the reference, KITTI/trainer.py:211) is
//     dx[ci](p)  = act_x'(x[ci](p)) * sum_c W1[c][ci] dz[c](p)                      -- head1x1_bwd_data_kernel
//     dW1[c][ci] = sum_p dz[c](p) x[ci](p),   db1[c] = sum_p dz[c](p)               -- head1x1_bwd_weight_kernel
// with dz the pre-activation gradient of the stacked 1x1 (what wmd_head3x3_bwd returns).  Both are skinny GEMMs (C = 32..256)
// bound by reading dz and x once and writing dx once; the generic convolution kernels run them at 14-37 TFLOP/s with 4-5x that
// traffic time (110 + 79 us at the finest level of BASELINE config 2).  Here: fp32 16x16x4 MFMA straight from global memory,
// pixels as the MFMA rows of the data gradient (a lane ends with 4 consecutive pixels of one channel: 16-byte gate loads /
// stores) and as the reduction index of the weight gradient (16-byte loads of 16 consecutive pixels per lane for both operands).
// (bodies take the block coordinates as arguments: they run as their own kernels, wmd_head_bwd1.hip, and as components of the
// merged second-stage launch of wmd_head_bwd, wmd_head_bwd.hip)
#pragma once
#include <algorithm>
#include <cstring>
#include "wmd_internal.h"

namespace wmd {

#ifndef WMD_F32X4_DEFINED
#define WMD_F32X4_DEFINED
typedef float f32x4 __attribute__((ext_vector_type(4)));
#endif

constexpr int H1_WW = 4;                      // wavefronts per block of the weight kernel (49 KB of LDS for the block reduction)
constexpr int H1_MAX_BLK = 512;               // block partials per channel-group pair
constexpr int H1_PART = 16 * 256 + 64;        // floats of one block's partial: 4 x 4 tiles of 16x16 + 64 bias sums
constexpr int H1_SMEM_FLOATS = (H1_WW - 1) * 16 * 64 * 4 + H1_WW * 64;   // block reduction of the weight kernel

struct Head1x1K {
    const float* dz;
    const float* x;
    const float* w1;
    float* dx;
    float* dw1;
    float* db1;
    float* partial;
    int B, HW, C, Ct;
    float dslope, delu;     // act'(x) = x > 0 ? 1 : dslope + delu * x
    int nblk, n_cgrp, n_igrp;   // weight kernel: 64-channel groups of dz / of x
};

// ---- data gradient: D[pixel 16][ci 16] += A[pixel][c] * B[c][ci], A = dz, B = W1 ------------------------------------------
// grid (pixel tiles, 64-wide groups of ci); a wave owns 64 pixels x <= 4 ci tiles; K = Ct walked four channels per MFMA.
// Row j of pixel group g is pixel 4 j + g of the tile: the four groups of a lane are FOUR CONSECUTIVE PIXELS, so the A operands
// of a K-step are one 16-byte load of dz (16 lanes = 256 contiguous bytes of one channel), and register i of the four groups'
// accumulators is four consecutive pixels (16 kq + 4 i + g) of output channel ci: 16-byte gate loads and stores.
__device__ __forceinline__ void head1x1_bwd_data_body(const Head1x1K& a, int bx, int by, int nbx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int HW = a.HW, T = (HW + 63) / 64;
    const int ci0 = by * 64;
    const int nct = min(4, (a.C - ci0 + 15) / 16);
    const bool vec = (HW & 3) == 0;
    const int KS = a.Ct / 4;
    for (int id = bx * 4 + wave; id < a.B * T; id += nbx * 4) {
        const int b = id / T, P0 = (id - b * T) * 64;
        f32x4 acc[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[g][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* dzp = a.dz + ((size_t)b * a.Ct + kq) * HW + P0 + 4 * j;     // + 4 s HW per K-step
        const float* wp = a.w1 + (size_t)kq * a.C + ci0 + j;                      // + 4 s C per K-step
        const bool full = vec && P0 + 64 <= HW;
        bool cok[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) cok[ct] = ct < nct && ci0 + ct * 16 + j < a.C;
        auto load_a = [&](int s, float (&A)[4]) {
            const float* p = dzp + (size_t)s * 4 * HW;
            if (full) {
                const float4 v = *reinterpret_cast<const float4*>(p);
                A[0] = v.x, A[1] = v.y, A[2] = v.z, A[3] = v.w;
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) A[g] = P0 + 4 * j + g < HW ? p[g] : 0.f;
            }
        };
        auto load_b = [&](int s, float (&Bv)[4]) {
            const float* p = wp + (size_t)s * 4 * a.C;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) Bv[ct] = cok[ct] ? p[ct * 16] : 0.f;
        };
        // two K-steps per iteration, the next pair's operands requested before this pair's MFMAs (Ct is a multiple of 8)
        float A0[4], A1[4], B0[4], B1[4];
        load_a(0, A0), load_a(1, A1), load_b(0, B0), load_b(1, B1);
        for (int s = 0; s < KS; s += 2) {
            float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a0[q] = A0[q], a1[q] = A1[q], b0[q] = B0[q], b1[q] = B1[q];
            if (s + 2 < KS) load_a(s + 2, A0), load_a(s + 3, A1), load_b(s + 2, B0), load_b(s + 3, B1);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (ct >= nct) break;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g], b0[ct], acc[g][ct], 0, 0, 0);
                    acc[g][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g], b1[ct], acc[g][ct], 0, 0, 0);
                }
            }
        }
        // lane (col = ci j; register i of group g = pixel 4 (4 kq + i) + g): float4 over g
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct >= nct) break;
            const int ci = ci0 + ct * 16 + j;
            if (ci >= a.C) continue;
            const size_t plane = ((size_t)b * a.C + ci) * HW;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int P = P0 + 16 * kq + 4 * i;
                if (P >= HW) continue;
                if (vec) {
                    const float4 m = *reinterpret_cast<const float4*>(a.x + plane + P);
                    float4 o;
                    o.x = acc[0][ct][i] * (m.x > 0.f ? 1.f : fmaf(a.delu, m.x, a.dslope));
                    o.y = acc[1][ct][i] * (m.y > 0.f ? 1.f : fmaf(a.delu, m.y, a.dslope));
                    o.z = acc[2][ct][i] * (m.z > 0.f ? 1.f : fmaf(a.delu, m.z, a.dslope));
                    o.w = acc[3][ct][i] * (m.w > 0.f ? 1.f : fmaf(a.delu, m.w, a.dslope));
                    *reinterpret_cast<float4*>(a.dx + plane + P) = o;
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (P + g < HW) {
                            const float m = a.x[plane + P + g];
                            a.dx[plane + P + g] = acc[g][ct][i] * (m > 0.f ? 1.f : fmaf(a.delu, m, a.dslope));
                        }
                }
            }
        }
    }
}

// ---- weight gradient: D[c 16][ci 16] += A[c][pixel] * B[pixel][ci] over the pixels -----------------------------------------
// grid (<= 256 pixel blocks, 64-channel group of dz x 64-channel group of x); K index kq of MFMA step s = pixel 16 kq + s of the
// wave's 64; a lane loads 16 consecutive pixels of ONE channel for both operands.  Accumulators (4 x 4 tiles) stay in registers
// over all tiles of the block; block partials + fixed-order reduce.
__device__ __forceinline__ void h1_load16(float (&v)[16], const float* p, bool ok, bool vec, int Pl, int HW) {
    if (vec && Pl + 16 <= HW) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 t = ok ? *reinterpret_cast<const float4*>(p + 4 * q4) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[4 * q4] = t.x, v[4 * q4 + 1] = t.y, v[4 * q4 + 2] = t.z, v[4 * q4 + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = (ok && Pl + s < HW) ? p[s] : 0.f;
    }
}

__device__ __forceinline__ void head1x1_bwd_weight_body(const Head1x1K& a, int bx, int by, int nbx, int nby, float* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int HW = a.HW, T = (HW + 63) / 64;
    const int cg = by / a.n_igrp, ig = by - cg * a.n_igrp;
    const int c0 = cg * 64, ci0 = ig * 64;
    const int nrt = min(4, (a.Ct - c0 + 15) / 16), nct = min(4, (a.C - ci0 + 15) / 16);
    const bool vec = (HW & 3) == 0;
    f32x4 acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};    // channel c0 + 16 rt + j, the lane's 16 pixels
    for (int id = bx * H1_WW + wave; id < a.B * T; id += nbx * H1_WW) {
        const int b = id / T, P0 = (id - b * T) * 64, Pl = P0 + 16 * kq;
        float xB[4][16];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int ci = ci0 + ct * 16 + j;
            h1_load16(xB[ct], a.x + ((size_t)b * a.C + min(ci, a.C - 1)) * HW + Pl, ct < nct && ci < a.C, vec, Pl, HW);
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            if (rt >= nrt) break;
            const int c = c0 + rt * 16 + j;
            float dA[16];
            h1_load16(dA, a.dz + ((size_t)b * a.Ct + min(c, a.Ct - 1)) * HW + Pl, c < a.Ct, vec, Pl, HW);
#pragma unroll
            for (int s = 0; s < 16; ++s) dbs[rt] += dA[s];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (ct >= nct) break;
#pragma unroll
                for (int s = 0; s < 16; ++s) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dA[s], xB[ct][s], acc[rt][ct], 0, 0, 0);
            }
        }
    }
    // bias sums: the four kq lanes of a channel, then the waves
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        dbs[rt] += __shfl_xor(dbs[rt], 16);
        dbs[rt] += __shfl_xor(dbs[rt], 32);
    }
    f32x4 (*red)[16][64] = reinterpret_cast<f32x4 (*)[16][64]>(smem);                 // [H1_WW - 1][16][64]
    float (*dbr)[64] = reinterpret_cast<float (*)[64]>(smem + (H1_WW - 1) * 16 * 64 * 4);   // [H1_WW][64]
    if (wave > 0) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) red[wave - 1][rt * 4 + ct][lane] = acc[rt][ct];
    }
    if (kq == 0) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) dbr[wave][rt * 16 + j] = dbs[rt];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.partial + ((size_t)bx * nby + by) * H1_PART;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                f32x4 v = acc[rt][ct];
                for (int w = 0; w < H1_WW - 1; ++w) v += red[w][rt * 4 + ct][lane];
                *reinterpret_cast<f32x4*>(out + ((rt * 4 + ct) * 64 + lane) * 4) = v;
            }
        float t = dbr[0][lane];
        for (int w = 1; w < H1_WW; ++w) t += dbr[w][lane];
        out[16 * 256 + lane] = t;
    }
}

// sums the block partials (sixteen threads per element, each a contiguous sixteenth of the blocks in order, combined pairwise
// in a fixed order) into dw1 [Ct, C] and db1 [Ct].  grid (channel-group pairs, 16-element chunks of 64*64 + 64)
__device__ __forceinline__ void head1x1_bwd_reduce_body(const Head1x1K& a, int bx, int by, int np) {
    const int cg = bx / a.n_igrp, ig = bx - cg * a.n_igrp;
    const int c0 = cg * 64, ci0 = ig * 64;
    const int e = by * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15;
    const bool is_db = e >= 64 * 64;
    const int r = is_db ? e - 64 * 64 : e >> 6, cl = is_db ? 0 : e & 63;     // row (dz channel) and column (x channel) inside the pair
    const bool live = e < 64 * 64 + 64 && c0 + r < a.Ct && (is_db ? ig == 0 : ci0 + cl < a.C);
    // D layout of tile (rt, ct): lane = (col = cl & 15) + 16 * ((r & 15) >> 2), register (r & 15) & 3
    const int src = is_db ? 16 * 256 + r : (((r >> 4) * 4 + (cl >> 4)) * 64 + (cl & 15) + 16 * ((r & 15) >> 2)) * 4 + (r & 3);
    const int per = (a.nblk + 15) / 16, b0 = q * per, b1 = min(b0 + per, a.nblk);
    float s = 0.f;
    if (live) {
        int blk = b0;
        for (; blk + 4 <= b1; blk += 4) {
            const float p0 = a.partial[((size_t)blk * np + bx) * H1_PART + src];
            const float p1 = a.partial[((size_t)(blk + 1) * np + bx) * H1_PART + src];
            const float p2 = a.partial[((size_t)(blk + 2) * np + bx) * H1_PART + src];
            const float p3 = a.partial[((size_t)(blk + 3) * np + bx) * H1_PART + src];
            s = (((s + p0) + p1) + p2) + p3;
        }
        for (; blk < b1; ++blk) s += a.partial[((size_t)blk * np + bx) * H1_PART + src];
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {      // (lower index) + (higher index) on both partners: a fixed tree
        const float o = __shfl_xor(s, m);
        s = q & m ? o + s : s + o;
    }
    if (live && q == 0) {
        if (is_db) a.db1[c0 + r] = s;
        else a.dw1[(size_t)(c0 + r) * a.C + ci0 + cl] = s;
    }
}


}  // namespace wmd
