// Wavelet-coefficient heads: 3x3 convolutions with 1..4 output channels (VALU, LDS-staged).
//
// Replaces the trailing Conv3x3(C,3|1) + Sigmoid of the `waveconv` heads and the
// 2^(s-1)*(sigmoid(+) - sigmoid(-)) combine of get_coefficients
// (KITTI/networks/decoders/depth_decoder.py:104-136), and the NYUv2 wave1_ll / wave{1,2,3}
// convolutions (NYUv2/networks/decoders/densedepth_decoder.py:106-115,122-141).
//
// With <= 4 output channels an MFMA tile would be >= 75 % padding, so this is a direct convolution on
// the vector ALU (same 157 TFLOP/s fp32 peak as the f32 MFMA):
//   * block = 128 threads = a 16 x 32 pixel tile; each thread owns 4 horizontally adjacent pixels, so one
//     ds_read_b128 + one ds_read_b64 per patch row feed 12 taps x 4 pixels (6 LDS instructions per 36*COUT FMAs)
//   * CK-channel halo patches (18 x 36, rows padded to a multiple of 4 floats so the vector reads are aligned)
//     staged global -> registers -> LDS, double buffered, one barrier per CK channels
//   * the CK x COUT x 9 filter taps of the chunk sit in LDS too and are read as wave-uniform (broadcast) b128
//   * epilogue: sigmoid / scale / (sigma+ - sigma-) and a float4 store per channel straight into the
//     [B,3,H,W] coefficient plane the IDWT reads.
#include <algorithm>
#include <cstdlib>
#include "wmd_internal.h"
#include "wmd_head_shiftsum.h"

namespace wmd {



constexpr int HT_H = 16, HT_W = 32, H_CK = 8, H_TT = 128;  // H_TT threads cover the tile once (4 px each)
constexpr int H_PH = HT_H + 2;        // 18 patch rows
constexpr int H_PWV = HT_W + 2;       // 34 valid patch columns
constexpr int H_PW = 36;              // row pitch (multiple of 4 floats)
constexpr int H_PS = H_PH * H_PW + 4; // channel stride 652 = 12 (mod 32): spreads channel planes over banks
// NG "channel groups" of H_TT threads share a tile: group g convolves channels g*CK/NG .. of every chunk and the
// partial sums are reduced through LDS.  NG = 4 (512 threads) gives the coarse levels (few pixels, up to 256
// channels) 4x the parallelism; staging is always cooperative over all NG*H_TT threads.

template <int COUT>
struct HeadSmem {
    float patch[2][H_CK * H_PS];
    float wgt[2][H_CK * COUT * 12];   // 9 taps padded to 12 per (ci, co): three aligned float4
    float dump[64];                    // target of the unconditional stores of lanes past the end of a patch
};

template <int COUT, int NG>
__device__ __forceinline__ void head_side(const float* __restrict__ x, size_t bstride, const float* __restrict__ wgt,
                                          int C, int c_begin, int c_end, int H, int W, int b, int y0, int x0,
                                          int pad_mode, HeadSmem<COUT>& sm, float (&acc)[COUT][4]) {
    constexpr int H_NT = H_TT * NG;
    constexpr int H_NPOS = (H_PH * H_PWV + H_NT - 1) / H_NT;
    constexpr int CPG = H_CK / NG;  // channels per group and chunk
    const int tid = threadIdx.x;
    const int grp = tid / H_TT, tt = tid % H_TT;
    const int ty = tt >> 3, tx = tt & 7;  // 16 rows x 8 column groups of 4 pixels
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * bstride;

    int off[H_NPOS];
    bool live[H_NPOS];
    int lpos[H_NPOS];
#pragma unroll
    for (int i = 0; i < H_NPOS; ++i) {
        const int p = min(tid + i * H_NT, H_PH * H_PWV - 1);
        const int py = p / H_PWV, px = p % H_PWV;
        int gy = y0 + py - 1, gx = x0 + px - 1;
        bool ok = (tid + i * H_NT) < H_PH * H_PWV;
        ok = pad_coord(gy, H, pad_mode) && ok;
        ok = pad_coord(gx, W, pad_mode) && ok;
        ok = ok && gy >= 0 && gx >= 0 && gy < H && gx < W;
        gy = min(max(gy, 0), H - 1);
        gx = min(max(gx, 0), W - 1);
        live[i] = ok;
        off[i] = gy * W + gx;
        lpos[i] = (tid + i * H_NT) < H_PH * H_PWV ? py * H_PW + px : -1;
    }
    // weights of a chunk: CK*COUT*9 values, thread t fetches elements t, t+128, ...
    constexpr int WN = H_CK * COUT * 9;
    constexpr int WPT = (WN + H_NT - 1) / H_NT;

    float sv[H_CK][H_NPOS];
    float wv[WPT];
    const int nchunks = (c_end - c_begin + H_CK - 1) / H_CK;  // this block's slice of the input channels

    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < H_CK; ++j) {
            const int ci = min(c_begin + chunk * H_CK + j, c_end - 1);
            const float* src = xb + (size_t)ci * plane;
#pragma unroll
            for (int i = 0; i < H_NPOS; ++i) sv[j][i] = src[off[i]];
        }
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int e = min(tid + k * H_NT, WN - 1);
            const int t = e % 9, co = (e / 9) % COUT, j = e / (9 * COUT);
            const int ci = min(c_begin + chunk * H_CK + j, c_end - 1);
            wv[k] = wgt[((size_t)co * C + ci) * 9 + t];
        }
    };
    auto stage_store = [&](int buf, int chunk) {
#pragma unroll
        for (int j = 0; j < H_CK; ++j) {
            const bool chan_ok = c_begin + chunk * H_CK + j < c_end;
#pragma unroll
            for (int i = 0; i < H_NPOS; ++i) {
                float* q = lpos[i] >= 0 ? &sm.patch[buf][j * H_PS + lpos[i]] : &sm.dump[tid & 63];
                *q = (live[i] && chan_ok) ? sv[j][i] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int e = tid + k * H_NT;
            const int ec = min(e, WN - 1);
            const int t = ec % 9, co = (ec / 9) % COUT, j = ec / (9 * COUT);
            float* q = e < WN ? &sm.wgt[buf][(j * COUT + co) * 12 + t] : &sm.dump[tid & 63];
            *q = (c_begin + chunk * H_CK + j < c_end) ? wv[k] : 0.f;
        }
    };

    __syncthreads();  // the previous side (or kernel prologue) is done with both buffers
    stage_load(0);
    stage_store(0, 0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        const int cn = min(c + 1, nchunks - 1);
        stage_load(cn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1  // one channel per trip: full unrolling hoists every LDS read and blows the register file
        for (int jj = 0; jj < CPG; ++jj) {
            const int j = grp * CPG + jj;
            const float* pt = &sm.patch[buf][j * H_PS + ty * H_PW + tx * 4];
            float v[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float4 lo = *reinterpret_cast<const float4*>(pt + r * H_PW);
                const float2 hi = *reinterpret_cast<const float2*>(pt + r * H_PW + 4);
                v[r][0] = lo.x; v[r][1] = lo.y; v[r][2] = lo.z; v[r][3] = lo.w; v[r][4] = hi.x; v[r][5] = hi.y;
            }
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float* wk = &sm.wgt[buf][(j * COUT + co) * 12];  // wave-uniform address: broadcast reads
                const float4 w0 = *reinterpret_cast<const float4*>(wk);
                const float4 w1 = *reinterpret_cast<const float4*>(wk + 4);
                const float w8 = wk[8];
                const float w[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w8};
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[co][q] = fmaf(w[t], v[t / 3][t % 3 + q], acc[co][q]);
            }
        }
        stage_store(buf ^ 1, cn);
        __syncthreads();
    }
    if (NG > 1) {
        // reduce the channel groups' partial sums into group 0 (the patch buffers are free now)
        float* red = &sm.patch[0][0];
        if (grp > 0) {
#pragma unroll
            for (int co = 0; co < COUT; ++co)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[((grp - 1) * COUT * 4 + co * 4 + q) * H_TT + tt] = acc[co][q];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int g = 1; g < NG; ++g)
#pragma unroll
                for (int co = 0; co < COUT; ++co)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[co][q] += red[((g - 1) * COUT * 4 + co * 4 + q) * H_TT + tt];
        }
    }
}

// Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2: block id -> tile such that an XCD owns one
// contiguous run of the tile sequence, so neighbouring tiles' halo columns / rows and shared 128-byte lines meet in ONE L2
// (counters: this kernel fetched 358 MB per launch at the finest level for 94 MB of input -- every tile's halo lines came from
// HBM again on another XCD -- and ran bandwidth-bound at 4.5 TB/s).  Same mapping as the convolution kernels (wmd_conv_common.h).
__device__ __forceinline__ int head_xcd_contiguous(int bid, int n) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int COUT, int NG>
__global__ __launch_bounds__(H_TT* NG) void head3x3_kernel(const wmd_head_args a, int tiles_x, int tiles_y, int csplit,
                                                            int cper, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) HeadSmem<COUT> sm;
    int t = head_xcd_contiguous(blockIdx.x, gridDim.x);
    const int tx_ = t % tiles_x;
    t /= tiles_x;
    const int ty_ = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty_ * HT_H, x0 = tx_ * HT_W;
    const int tt_ = threadIdx.x % H_TT;
    const int oy = y0 + (tt_ >> 3), ox = x0 + (tt_ & 7) * 4;

    float accp[COUT][4], accn[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // the bias enters once: through group 0 of an unsplit block, or in the finalize pass
            const bool g0 = threadIdx.x < H_TT && csplit == 1;
            accp[co][q] = (g0 && a.bias_p) ? a.bias_p[co] : 0.f;
            accn[co][q] = (g0 && a.mode == 2 && a.bias_n) ? a.bias_n[co] : 0.f;
        }
    const size_t plane_ = (size_t)a.H * a.W;
    const int cs = blockIdx.y;
    const int c_begin = cs * cper, c_end = min(a.C, c_begin + cper);
    head_side<COUT, NG>(a.xp, a.xp_bstride ? a.xp_bstride : a.C * plane_, a.wgt_p, a.C, c_begin, c_end, a.H, a.W, b, y0,
                        x0, a.pad_mode, sm, accp);
    if (a.mode == 2)
        head_side<COUT, NG>(a.xn, a.xn_bstride ? a.xn_bstride : a.C * plane_, a.wgt_n, a.C, c_begin, c_end, a.H, a.W, b,
                            y0, x0, a.pad_mode, sm, accn);

    if (threadIdx.x >= H_TT || oy >= a.H || ox >= a.W) return;
    const size_t plane = (size_t)a.H * a.W;
    if (csplit > 1) {
        // raw partial sums [cs][b][side][co][H][W]; bias, sigmoid and the combine happen in head_finalize_kernel
        float* pp = partial + (((size_t)cs * a.B + b) * 2 * COUT) * plane + (size_t)oy * a.W + ox;
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ox + q < a.W) {
                    pp[(size_t)co * plane + q] = accp[co][q];
                    if (a.mode == 2) pp[(size_t)(COUT + co) * plane + q] = accn[co][q];
                }
        return;
    }
    const size_t o = (size_t)b * COUT * plane + (size_t)oy * a.W + ox;
    const bool vec = (ox + 3 < a.W) && ((a.W & 3) == 0);
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float r[4], sp[4], sn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (a.mode == 0) {
                r[q] = a.scale * accp[co][q];
            } else {
                sp[q] = 1.f / (1.f + expf(-accp[co][q]));
                if (a.mode == 1) {
                    r[q] = a.scale * sp[q];
                } else {
                    sn[q] = 1.f / (1.f + expf(-accn[co][q]));
                    // reference order (depth_decoder.py:134-135): 2^(s-1)*sig(+) - 2^(s-1)*sig(-)
                    r[q] = a.scale * sp[q] - a.scale * sn[q];
                }
            }
        }
        float* yo = a.y + o + co * plane;
        if (vec) {
            *reinterpret_cast<float4*>(yo) = make_float4(r[0], r[1], r[2], r[3]);
            if (a.mode >= 1 && a.sig_p) *reinterpret_cast<float4*>(a.sig_p + o + co * plane) = make_float4(sp[0], sp[1], sp[2], sp[3]);
            if (a.mode == 2 && a.sig_n) *reinterpret_cast<float4*>(a.sig_n + o + co * plane) = make_float4(sn[0], sn[1], sn[2], sn[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ox + q < a.W) {
                    yo[q] = r[q];
                    if (a.mode >= 1 && a.sig_p) a.sig_p[o + co * plane + q] = sp[q];
                    if (a.mode == 2 && a.sig_n) a.sig_n[o + co * plane + q] = sn[q];
                }
        }
    }
}

// second stage of a channel-split head: sum the slices, add the bias, apply sigmoid / scale / (s+ - s-)
__global__ void head_finalize_kernel(const wmd_head_args a, const float* __restrict__ partial, int csplit) {
    const size_t plane = (size_t)a.H * a.W;
    const size_t n = (size_t)a.B * a.Cout * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i % plane;
        const int co = (i / plane) % a.Cout;
        const int b = i / (plane * a.Cout);
        float p = a.bias_p ? a.bias_p[co] : 0.f, q = (a.mode == 2 && a.bias_n) ? a.bias_n[co] : 0.f;
        for (int cs = 0; cs < csplit; ++cs) {
            const float* pp = partial + (((size_t)cs * a.B + b) * 2 * a.Cout) * plane + pix;
            p += pp[(size_t)co * plane];
            if (a.mode == 2) q += pp[(size_t)(a.Cout + co) * plane];
        }
        float r;
        if (a.mode == 0) {
            r = a.scale * p;
        } else {
            const float sp = 1.f / (1.f + expf(-p));
            if (a.sig_p) a.sig_p[i] = sp;
            if (a.mode == 1) {
                r = a.scale * sp;
            } else {
                const float sn = 1.f / (1.f + expf(-q));
                if (a.sig_n) a.sig_n[i] = sn;
                r = a.scale * sp - a.scale * sn;
            }
        }
        a.y[i] = r;
    }
}

// Completes the fused head: y[co][p] = scale*sig(b+[co] + sum_tap t+[co*9+tap][pad(p+d_tap)]) - scale*sig(same for -),
// and optionally the Haar synthesis of (yl, y) -> out / disp.  One thread per coefficient pixel; HBM-bound
// (54 planes read about once through the caches, 3 + 4 (+4) values written).
__global__ void head_shiftsum_kernel(const wmd_head_shiftsum_args a) {
    const int H = a.H, W = a.W;
    const size_t plane = (size_t)H * W;
    const size_t total = (size_t)a.B * plane;
    // (the loop bound is wave-uniform: the tail lanes of the last wavefront run along masked out, so that the wavefront-wide
    //  ballots / shuffles below see every lane)
    const size_t total_w = (total + 63) / 64 * 64;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total_w; i0 += (size_t)gridDim.x * blockDim.x) {
        const bool live = i0 < total;
        const size_t i = live ? i0 : total - 1;
        const int x = i % W;
        const int y = (i / W) % H;
        const size_t b = i / plane;
        // all 54 loads are unconditional (clamped offset, 0/1 weight) so that they are issued back to back: with the tap
        // test around them the compiler serialises the round trips and this small kernel took 14 us per level
        int off[9];
        float okf[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int gy = y + t / 3 - 1, gx = x + t % 3 - 1;
            const bool ok = pad_coord(gy, H, a.pad_mode) & pad_coord(gx, W, a.pad_mode);
            gy = min(max(gy, 0), H - 1);
            gx = min(max(gx, 0), W - 1);
            off[t] = gy * W + gx;
            okf[t] = ok ? 1.f : 0.f;
        }
        const float* tb = a.t + b * (a.yl_out ? 81 : 54) * plane;
        // block-sparse levels: a pixel outside the wavelet mask has yh = 0 whatever its taps hold, and a wavefront without a
        // single mask pixel skips the 54 gathers altogether (its tap-partial planes may never have been written:
        // wmd_head_fused_args.run_mask)
        const bool in_mask = !a.yh_mask || a.yh_mask[b * plane + (size_t)y * W + x] != 0;
        const bool wave_any = __builtin_amdgcn_ballot_w64(in_mask && live) != 0;
        float yh[3] = {0.f, 0.f, 0.f};
        if (wave_any) {
            float vp[27], vn[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                vp[k] = tb[(size_t)k * plane + off[k % 9]];
                vn[k] = tb[(size_t)(27 + k) * plane + off[k % 9]];
            }
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                float sp = a.bias_p ? a.bias_p[co] : 0.f, sn = a.bias_n ? a.bias_n[co] : 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    sp += okf[t] * vp[co * 9 + t];
                    sn += okf[t] * vn[co * 9 + t];
                }
                const float a1 = fast_sigmoid(sp), a2 = fast_sigmoid(sn);
                yh[co] = in_mask ? a.scale * a1 - a.scale * a2 : 0.f;
                if (a.sig_p && live) {   // training forward: what the heads' backward multiplies by
                    a.sig_p[(b * 3 + co) * plane + (size_t)y * W + x] = a1;
                    a.sig_n[(b * 3 + co) * plane + (size_t)y * W + x] = a2;
                }
            }
        }
        if (live) {
#pragma unroll
            for (int co = 0; co < 3; ++co) a.yh[(b * 3 + co) * plane + (size_t)y * W + x] = yh[co];
        }
        float l = 0.f;
        if (a.yl_out) {   // the low-pass head of the coarsest level: third chain of the fused launch, rows 54..62
            float sl = a.bias_ll ? a.bias_ll[0] : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) sl += okf[t] * tb[(size_t)(54 + t) * plane + off[t]];
            const float sgl = fast_sigmoid(sl);
            l = a.scale_ll * sgl;
            if (live) a.yl_out[i] = l;
            if (live && a.sig_ll) a.sig_ll[i] = sgl;     // (what the backward multiplies by: the value the forward used)
        } else if (a.yl) {
            l = a.yl[i];
        }
        if ((a.yl || a.yl_out) && a.out) {
            float v[4] = {(l + yh[0] + yh[1] + yh[2]) * 0.5f, (l + yh[0] - yh[1] - yh[2]) * 0.5f,
                          (l - yh[0] + yh[1] - yh[2]) * 0.5f, (l - yh[0] - yh[1] + yh[2]) * 0.5f};
            if (a.range_keys) {
                // the range (min, max) of every frame's new low-pass plane for the next level's threshold: wavefront reduction,
                // one atomic pair per wavefront (a wavefront that straddles two frames lets every lane speak for itself)
                float lo = live ? fminf(fminf(v[0], v[1]), fminf(v[2], v[3])) : INFINITY;
                float hi = live ? fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) : -INFINITY;
                const int b0 = __builtin_amdgcn_readfirstlane((int)b);
                const bool uniform = __builtin_amdgcn_ballot_w64((int)b != b0) == 0;
                if (uniform) {
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        lo = fminf(lo, __shfl_xor(lo, o));
                        hi = fmaxf(hi, __shfl_xor(hi, o));
                    }
                }
                if ((uniform ? (threadIdx.x & 63) == 0 : live) && lo <= hi) {
                    const unsigned ul = __float_as_uint(lo), uh = __float_as_uint(hi);
                    atomicMin(&a.range_keys[2 * b], (ul & 0x80000000u) ? ~ul : (ul | 0x80000000u));
                    atomicMax(&a.range_keys[2 * b + 1], (uh & 0x80000000u) ? ~uh : (uh | 0x80000000u));
                }
            }
            if (!live) continue;
            const size_t dst = b * 4 * plane + (size_t)(2 * y) * (2 * W) + 2 * x;
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

// ---- the shift-sums of up to three consecutive levels in ONE launch (round 5, dense inference) ---------------------------------
// Level k's synthesis output IS level k+1's low-pass input, pixel for pixel (out_k [2h,2w] = yl_{k+1}), so once the chained GEMM
// kernels of all levels have written their tap-partial planes the three completions have no reason to be three graph nodes of
// 7-10 us each (every one a launch of a few hundred small workgroups on its latency floor): a block takes a 4 x 4 tile of the
// coarsest level, 16 threads finish it (nine-tap gathers, bias, sigmoid, combine, Haar butterfly) and leave the 8 x 8 low-pass
// tile of the next level in LDS, 64 threads finish that one, 256 the third.  Same per-pixel arithmetic as head_shiftsum_kernel
// (shared body below), same outputs per level.
struct ShiftsumChainArgs {
    wmd_head_shiftsum_args lv[3];
    int n;
    int by0, bx0;   // tile of the coarsest level (by0 * bx0 == 16)
};


__global__ __launch_bounds__(256) void head_shiftsum_chain_kernel(const ShiftsumChainArgs c) {
    __shared__ float low[2][256];     // low-pass tiles handed from level k to level k + 1
    // tile of the coarsest level: by0 x bx0 pixels (16 of them), doubling per level up to 256 at the third.  Round 6: 2 x 8 instead
    // of 4 x 4 -- the finest level's 8 x 32 tile reads and writes whole 128-byte rows where the 16 x 16 one touched 64-byte halves
    const int by0 = c.by0, bx0 = c.bx0;
    const int H0 = c.lv[0].H, W0 = c.lv[0].W;
    const int tiles_x = (W0 + bx0 - 1) / bx0, tiles_y = (H0 + by0 - 1) / by0;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const size_t b = t / tiles_y;
    const int tid = threadIdx.x;
    for (int k = 0; k < c.n; ++k) {
        const int sy = by0 << k, sx = bx0 << k;        // the tile at level k
        if (tid < sy * sx) {
            const int py = tid / sx, px = tid % sx;
            const int y = ty * sy + py, x = tx * sx + px;
            if (y < c.lv[k].H && x < c.lv[k].W) {
                float v[4];
                shiftsum_pixel(c.lv[k], b, y, x, k > 0, k > 0 ? low[(k - 1) & 1][py * sx + px] : 0.f, v);
                if (k + 1 < c.n) {
                    float* nl = low[k & 1];
                    const int ns = sx * 2;
                    nl[(2 * py) * ns + 2 * px] = v[0];
                    nl[(2 * py) * ns + 2 * px + 1] = v[1];
                    nl[(2 * py + 1) * ns + 2 * px] = v[2];
                    nl[(2 * py + 1) * ns + 2 * px + 1] = v[3];
                }
            }
        }
        __syncthreads();
    }
}

static void head_plan(const wmd_head_args* g, int* tiles_x, int* tiles_y, int* csplit, int* cper) {
    *tiles_x = (g->W + HT_W - 1) / HT_W;
    *tiles_y = (g->H + HT_H - 1) / HT_H;
    const long tiles = (long)g->B * *tiles_x * *tiles_y;
    // few tiles (coarse pyramid levels, up to 256 channels each): slice the channel loop over blocks so that
    // ~3 blocks per CU are in flight; a slice is at least two 8-channel chunks
    int cs = (int)std::min<long>((3L * kNumCU + tiles - 1) / tiles, std::max(1, g->C / 16));
    if (const char* e = getenv("WMD_HEAD_CSPLIT")) cs = std::max(1, atoi(e));
    cs = std::max(1, std::min(cs, (g->C + H_CK - 1) / H_CK));
    int per = (g->C + cs - 1) / cs;
    per = ((per + H_CK - 1) / H_CK) * H_CK;
    *cper = per;
    *csplit = (g->C + per - 1) / per;
}

}  // namespace wmd

using namespace wmd;

extern "C" size_t wmd_head3x3_workspace_floats(const wmd_head_args* g) {
    if (!g || g->B <= 0 || g->C <= 0 || g->Cout <= 0) return 0;
    int tx, ty, cs, per;
    head_plan(g, &tx, &ty, &cs, &per);
    return cs > 1 ? (size_t)cs * g->B * 2 * g->Cout * g->H * g->W : 0;
}

extern "C" int wmd_head3x3_fwd(const wmd_head_args* g, void* stream) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_fwd: null args");
    if (!g->xp || !g->wgt_p || !g->y) return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_fwd: null tensor pointer");
    if (g->mode < 0 || g->mode > 2) return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_fwd: mode=%d", g->mode);
    if (g->mode == 2 && (!g->xn || !g->wgt_n)) return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_fwd: mode 2 needs xn/wgt_n");
    if (g->B <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0)
        return fail(WMD_ERR_BAD_SHAPE, "wmd_head3x3_fwd: B=%d H=%d W=%d C=%d", g->B, g->H, g->W, g->C);
    if (g->Cout < 1 || g->Cout > 4) return fail(WMD_ERR_UNSUPPORTED, "wmd_head3x3_fwd: Cout=%d (1..4)", g->Cout);
    if (g->pad_mode < 0 || g->pad_mode > 2) return fail(WMD_ERR_BAD_ARG, "wmd_head3x3_fwd: pad_mode=%d", g->pad_mode);
    if (g->pad_mode == WMD_PAD_REFLECT && (g->H < 2 || g->W < 2))
        return fail(WMD_ERR_BAD_SHAPE, "wmd_head3x3_fwd: reflect padding needs H,W >= 2");
    int tiles_x, tiles_y, csplit, cper;
    head_plan(g, &tiles_x, &tiles_y, &csplit, &cper);
    if (csplit > 1 && (!g->workspace || g->workspace_floats < (size_t)csplit * g->B * 2 * g->Cout * g->H * g->W)) {
        csplit = 1;  // no (or too small a) workspace: run unsplit
        cper = ((g->C + H_CK - 1) / H_CK) * H_CK;
    }
    dim3 grid((unsigned)((size_t)g->B * tiles_x * tiles_y), (unsigned)csplit);
    hipStream_t s = (hipStream_t)stream;
    const double pix = (double)g->B * g->H * g->W, sides = g->mode == 2 ? 2.0 : 1.0;
    ProfScope prof("head3x3_kernel", sides * 18.0 * g->C * g->Cout * pix, 4.0 * pix * (sides * g->C + g->Cout), s);
    // few tiles (coarse pyramid levels): 4 channel groups per tile; otherwise 2
    bool wide = (size_t)g->B * tiles_x * tiles_y * csplit < 2 * (size_t)kNumCU;
    if (const char* e = getenv("WMD_HEAD_NG")) wide = atoi(e) == 4;
#define WMD_HEAD_LAUNCH(CO)                                                                                   \
    if (wide) hipLaunchKernelGGL((head3x3_kernel<CO, 4>), grid, dim3(H_TT * 4), 0, s, *g, tiles_x, tiles_y, csplit, cper, g->workspace); \
    else hipLaunchKernelGGL((head3x3_kernel<CO, 2>), grid, dim3(H_TT * 2), 0, s, *g, tiles_x, tiles_y, csplit, cper, g->workspace)
    switch (g->Cout) {
        case 1: WMD_HEAD_LAUNCH(1); break;
        case 2: WMD_HEAD_LAUNCH(2); break;
        case 3: WMD_HEAD_LAUNCH(3); break;
        default: WMD_HEAD_LAUNCH(4); break;
    }
#undef WMD_HEAD_LAUNCH
    int st = check_launch("head3x3_kernel");
    if (st || csplit == 1) return st;
    const size_t n = (size_t)g->B * g->Cout * g->H * g->W;
    hipLaunchKernelGGL(head_finalize_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, s, *g,
                       g->workspace, csplit);
    return check_launch("head_finalize_kernel");
}

extern "C" int wmd_head_shiftsum_chain_fwd(const wmd_head_shiftsum_args* levels, int n_levels, void* stream) {
    if (!levels || n_levels < 1 || n_levels > 3) return fail(WMD_ERR_BAD_ARG, "wmd_head_shiftsum_chain_fwd: 1..3 levels (got %d)", n_levels);
    ShiftsumChainArgs c;
    c.n = n_levels;
    for (int k = 0; k < n_levels; ++k) {
        const wmd_head_shiftsum_args& g = levels[k];
        if (!g.t || !g.yh || !g.out) return fail(WMD_ERR_BAD_ARG, "wmd_head_shiftsum_chain_fwd: level %d: t, yh and out are required", k);
        if (g.B != levels[0].B || g.H != (levels[0].H << k) || g.W != (levels[0].W << k))
            return fail(WMD_ERR_BAD_SHAPE, "wmd_head_shiftsum_chain_fwd: level %d is %dx%dx%d, expected %dx%dx%d (each level doubles the one before)",
                        k, g.B, g.H, g.W, levels[0].B, levels[0].H << k, levels[0].W << k);
        if (g.pad_mode < 0 || g.pad_mode > 2) return fail(WMD_ERR_BAD_ARG, "wmd_head_shiftsum_chain_fwd: pad_mode=%d", g.pad_mode);
        if (g.pad_mode == WMD_PAD_REFLECT && (g.H < 2 || g.W < 2)) return fail(WMD_ERR_BAD_SHAPE, "wmd_head_shiftsum_chain_fwd: reflect padding needs H,W >= 2");
        if (g.yh_mask || g.range_keys || g.sig_p || g.sig_n || g.sig_ll)
            return fail(WMD_ERR_UNSUPPORTED, "wmd_head_shiftsum_chain_fwd: dense inference only (no yh_mask / range_keys / sigmoid outputs)");
        if (k == 0 ? ((g.yl != nullptr) == (g.yl_out != nullptr)) : (g.yl != nullptr || g.yl_out != nullptr))
            return fail(WMD_ERR_BAD_ARG, "wmd_head_shiftsum_chain_fwd: the first level takes exactly one of yl / yl_out, the others take the chain's low-pass");
        c.lv[k] = g;
    }
    for (int k = n_levels; k < 3; ++k) c.lv[k] = levels[0];
    static const int square = [] {
        const char* e = getenv("WMD_SHIFTSUM_CHAIN_SQUARE");
        return e ? atoi(e) : 0;
    }();
    c.by0 = square ? 4 : 2;
    c.bx0 = square ? 4 : 8;
    const int tiles = ((levels[0].W + c.bx0 - 1) / c.bx0) * ((levels[0].H + c.by0 - 1) / c.by0);
    hipStream_t s = (hipStream_t)stream;
    double n = 0;
    for (int k = 0; k < n_levels; ++k) n += (double)levels[k].B * levels[k].H * levels[k].W;
    ProfScope prof("head_shiftsum_chain_kernel", 60.0 * n, 4.0 * n * 66, s);
    hipLaunchKernelGGL(head_shiftsum_chain_kernel, dim3((unsigned)(levels[0].B * tiles)), dim3(256), 0, s, c);
    return check_launch("head_shiftsum_chain_kernel");
}

extern "C" int wmd_head_shiftsum_fwd(const wmd_head_shiftsum_args* g, void* stream) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "wmd_head_shiftsum_fwd: null args");
    if (!g->t || !g->yh) return fail(WMD_ERR_BAD_ARG, "wmd_head_shiftsum_fwd: null tensor pointer");
    if (g->B <= 0 || g->H <= 0 || g->W <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_head_shiftsum_fwd: B=%d H=%d W=%d", g->B, g->H, g->W);
    if (g->pad_mode < 0 || g->pad_mode > 2) return fail(WMD_ERR_BAD_ARG, "wmd_head_shiftsum_fwd: pad_mode=%d", g->pad_mode);
    if (g->pad_mode == WMD_PAD_REFLECT && (g->H < 2 || g->W < 2))
        return fail(WMD_ERR_BAD_SHAPE, "wmd_head_shiftsum_fwd: reflect padding needs H,W >= 2");
    if (g->yl_out && g->yl) return fail(WMD_ERR_BAD_ARG, "wmd_head_shiftsum_fwd: yl_out (low-pass head completed here) excludes yl");
    if ((g->out != nullptr) != (g->yl != nullptr || g->yl_out != nullptr))
        return fail(WMD_ERR_BAD_ARG, "wmd_head_shiftsum_fwd: yl (or yl_out) and out go together");
    if ((g->sig_p != nullptr) != (g->sig_n != nullptr) || (g->sig_ll && !g->yl_out) || (g->sig_p && g->yh_mask))
        return fail(WMD_ERR_BAD_ARG, "wmd_head_shiftsum_fwd: sig_p and sig_n go together, sig_ll needs yl_out, neither takes a yh_mask");
    const size_t n = (size_t)g->B * g->H * g->W;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("head_shiftsum_kernel", 60.0 * n, 4.0 * n * ((g->yl_out ? 64 : 54) + 3 + (g->out ? (g->disp ? 9 : 5) : 0)), s);
    const unsigned threads = n < 65536 ? 64 : 256;   // coarse levels: one wavefront per block spreads the few pixels over the CUs
    hipLaunchKernelGGL(head_shiftsum_kernel, dim3((unsigned)std::min<size_t>((n + threads - 1) / threads, (size_t)kNumCU * 16)), dim3(threads),
                       0, s, *g);
    return check_launch("head_shiftsum_kernel");
}
