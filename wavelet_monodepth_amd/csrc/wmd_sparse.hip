// Threshold-gated sparse decoder path (batch 1) for gfx950.
//
// Reference: SparseDepthWaveProgressiveDecoder.forward (KITTI/networks/decoders/depth_decoder.py:292-428),
// SparseDecoderWave.forward (NYUv2/networks/decoders/densedepth_decoder.py:271-409) and the sparse_* helpers
// (KITTI/layers.py:337-507).  There every level is a storm of boolean-mask gathers, index_puts and small
// matmuls with a host sync (`.sum()` -> `arange(numel)`) per index map.  Here:
//   minmax_kernel            global min/max of the LL plane                      (depth_decoder.py:308)
//   mask_threshold_kernel    max_b |yh_b| > ratio * range                        (:308-309)
//   mask_dilate_multi_kernel every MaxPool2d(3|5)(upsample?) variant at once     (:311-319)
//   mask_compact_multi_kernel raster-order stream compaction: wavefront __ballot + popcount prefix sums,
//                            one workgroup per mask, count left on the device     (layers.py:371-389)
//   sparse_conv_kernel       gather-GEMM on fp32 MFMA over the compacted active pixels, fused
//                            select/upsample/concat/pad/bias/activation/scatter   (layers.py:337-507)
// Activations stay dense and zero-initialised; "not in the input mask => reads 0" is a mask test after the
// coordinate padding, which is exactly what padding the index map does in the reference (layers.py:444).
#include <algorithm>
#include "wmd_internal.h"

namespace wmd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void minmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out2) {
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        const float v = x[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    __shared__ float slo[16], shi[16];
    if ((threadIdx.x & 63) == 0) {
        slo[threadIdx.x >> 6] = lo;
        shi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            lo = fminf(lo, slo[w]);
            hi = fmaxf(hi, shi[w]);
        }
        out2[0] = lo;
        out2[1] = hi;
    }
}

__global__ void mask_threshold_kernel(const float* __restrict__ yh, const float* __restrict__ minmax, float ratio,
                                      uint8_t* __restrict__ mask, int npix) {
    const float thr = (minmax[1] - minmax[0]) * ratio;  // fp32, like the reference's 0-dim tensor arithmetic
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const float m = fmaxf(fmaxf(fabsf(yh[i]), fabsf(yh[npix + i])), fabsf(yh[2 * npix + i]));
        mask[i] = m > thr ? 1 : 0;
    }
}

struct DilateKArgs {
    wmd_dilate_spec s[8];
};

__global__ void mask_dilate_multi_kernel(const uint8_t* __restrict__ mask, int h, int w, const DilateKArgs a) {
    const wmd_dilate_spec sp = a.s[blockIdx.y];
    const int H = h * sp.up, W = w * sp.up, r = sp.radius;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i % W;
        uint8_t v = 0;
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;  // MaxPool2d pads with -inf: out-of-range taps never win
            for (int dx = -r; dx <= r; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                v |= mask[(yy / sp.up) * w + xx / sp.up];
            }
        }
        sp.out[i] = v;
    }
}

struct CompactKArgs {
    wmd_compact_spec s[8];
};

// One workgroup (16 wavefronts) per mask.  Per 1024-pixel chunk: every wavefront takes a 64-bit __ballot of its
// flags; a lane's slot is popcount(ballot & lanes-below); wavefront totals are scanned through LDS.
__global__ __launch_bounds__(1024) void mask_compact_multi_kernel(const CompactKArgs a) {
    const wmd_compact_spec sp = a.s[blockIdx.x];
    __shared__ int wave_tot[16];
    __shared__ int running;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < sp.npix; base += 1024) {
        const int i = base + threadIdx.x;
        const bool flag = i < sp.npix && sp.mask[i] != 0;
        const unsigned long long bal = __ballot(flag);
        const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = running;
        for (int wv = 0; wv < wave; ++wv) off += wave_tot[wv];
        if (flag) sp.coords[off + prefix] = i;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int wv = 0; wv < 16; ++wv) tot += wave_tot[wv];
            running += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *sp.nnz = running;
}

// ------------------------------------------------------------------------------------------------
struct SparseKArgs {
    wmd_sparse_conv_args g;
    int nci4, ncot, W1;
    size_t plane, plane1;
};

// WK wavefronts share one (16-pixel tile, MR out-channel tiles) task and split the input-channel loop between
// them (the chain is latency-bound: a single wave would walk all Cin*9 gathers serially); partial accumulators
// are combined through LDS by wave 0.
template <int MR, int TAPS, bool DUAL, int WK>
__global__ __launch_bounds__(64 * WK) void sparse_conv_kernel(const SparseKArgs a) {
    const wmd_sparse_conv_args& g = a.g;
    const int nnz = min(*g.out_nnz, g.max_out);
    const int tile = blockIdx.x;
    if (tile * 16 >= nnz) return;
    const int lane = threadIdx.x & 63;
    const int wk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int pidx = tile * 16 + j;
    const bool px_ok = pidx < nnz;
    const int p = g.out_coords[min(pidx, nnz - 1)];
    const int oy = p / g.W, ox = p % g.W;
    const int Cin = g.C1 + g.C2;

    // neighbour offsets through the coordinate padding + input-mask test (layers.py:439-453)
    int o1[TAPS], o2[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        int gy = oy + (TAPS == 9 ? t / 3 - 1 : 0), gx = ox + (TAPS == 9 ? t % 3 - 1 : 0);
        bool ok = true;
        if (TAPS == 9) {
            ok = pad_coord(gy, g.H, g.pad_mode) && ok;
            ok = pad_coord(gx, g.W, g.pad_mode) && ok;
        }
        gy = min(max(gy, 0), g.H - 1);
        gx = min(max(gx, 0), g.W - 1);
        if (g.in_mask) ok = ok && g.in_mask[gy * g.W + gx] != 0;
        o2[t] = ok ? gy * g.W + gx : -1;
        o1[t] = ok ? (gy / g.up1) * a.W1 + gx / g.up1 : -1;
    }

    f32x4 acc[MR], acc2[DUAL ? MR : 1];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (DUAL) {
#pragma unroll
        for (int m = 0; m < MR; ++m) acc2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int cot0 = blockIdx.y * MR;
    const float* wa[MR];
    const float* wa2[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int cot = min(cot0 + m, a.ncot - 1);
        wa[m] = g.wp + (size_t)cot * a.nci4 * (TAPS * 64) + lane;
        wa2[m] = DUAL ? g.wp2 + (size_t)cot * a.nci4 * (TAPS * 64) + lane : nullptr;
    }

    const int nci4 = (Cin + 3) / 4;
#pragma unroll 2
    for (int ci4 = wk; ci4 < nci4; ci4 += WK) {
        const int ci = ci4 * 4 + kq;
        const bool from1 = ci < g.C1;
        const bool ch_ok = ci < Cin;
        const int c1 = min(ci, g.C1 - 1), c2 = min(max(ci - g.C1, 0), max(g.C2 - 1, 0));
        const float* s1 = g.x1 + (size_t)(g.c1_off + c1) * a.plane1;
        const float* s1b = DUAL ? g.x1 + (size_t)(g.c1_off2 + c1) * a.plane1 : nullptr;
        const float* s2 = g.x2 ? g.x2 + (size_t)c2 * a.plane : s1;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            // unconditional loads from clamped offsets, zeroing by select afterwards
            const int q1 = max(o1[t], 0), q2 = max(o2[t], 0);
            const float v1 = s1[q1];
            const float v2 = g.x2 ? s2[q2] : 0.f;
            const bool live = ch_ok && o2[t] >= 0;
            const float b = live ? (from1 ? v1 : v2) : 0.f;
            float b2 = 0.f;
            if (DUAL) {
                const float v1b = s1b[q1];
                b2 = (live && from1) ? v1b : 0.f;
            }
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const float af = wa[m][(size_t)(ci4 * TAPS + t) * 64];
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b, acc[m], 0, 0, 0);
                if (DUAL) {
                    const float af2 = wa2[m][(size_t)(ci4 * TAPS + t) * 64];
                    acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af2, b2, acc2[m], 0, 0, 0);
                }
            }
        }
    }

    if (WK > 1) {
        __shared__ f32x4 red[WK > 1 ? WK - 1 : 1][DUAL ? 2 * MR : MR][64];
        if (wk > 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                red[wk - 1][m][lane] = acc[m];
                if (DUAL) red[wk - 1][MR + m][lane] = acc2[m];
            }
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int w = 0; w < WK - 1; ++w)
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                acc[m] += red[w][m][lane];
                if (DUAL) acc2[m] += red[w][MR + m][lane];
            }
    }
    if (!px_ok) return;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = (cot0 + m) * 16 + kq * 4 + r;
            if (co < g.Cout) {
                float v = acc[m][r] + (g.bias ? g.bias[co] : 0.f);
                v = g.out_scale * act_apply(v, g.act, g.slope);
                if (DUAL) {
                    float u = acc2[m][r] + (g.bias2 ? g.bias2[co] : 0.f);
                    v = v - g.out_scale * act_apply(u, g.act, g.slope);
                }
                g.y[(size_t)co * a.plane + p] = v;
            }
        }
}

}  // namespace wmd

using namespace wmd;

extern "C" int wmd_minmax(const float* x, size_t n, float* out2, void* stream) {
    if (!x || !out2) return fail(WMD_ERR_BAD_ARG, "wmd_minmax: null pointer");
    if (n == 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_minmax: empty input (torch.max raises too)");
    ProfScope prof("minmax_kernel", (double)n, 4.0 * n, (hipStream_t)stream);
    hipLaunchKernelGGL(minmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out2);
    return check_launch("minmax_kernel");
}

extern "C" int wmd_mask_threshold(const float* yh, const float* minmax, float thresh_ratio, uint8_t* mask, int h, int w,
                                  void* stream) {
    if (!yh || !minmax || !mask) return fail(WMD_ERR_BAD_ARG, "wmd_mask_threshold: null pointer");
    if (h <= 0 || w <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_threshold: h=%d w=%d", h, w);
    const int npix = h * w;
    ProfScope prof("mask_threshold_kernel", 3.0 * npix, 13.0 * npix, (hipStream_t)stream);
    hipLaunchKernelGGL(mask_threshold_kernel, dim3(std::min((npix + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                       yh, minmax, thresh_ratio, mask, npix);
    return check_launch("mask_threshold_kernel");
}

extern "C" int wmd_mask_dilate_multi(const uint8_t* mask, int h, int w, const wmd_dilate_spec* specs, int n, void* stream) {
    if (!mask || !specs) return fail(WMD_ERR_BAD_ARG, "wmd_mask_dilate_multi: null pointer");
    if (h <= 0 || w <= 0 || n <= 0 || n > 8) return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_dilate_multi: h=%d w=%d n=%d", h, w, n);
    DilateKArgs a;
    int maxpix = 0;
    for (int i = 0; i < n; ++i) {
        if (!specs[i].out || (specs[i].up != 1 && specs[i].up != 2) || specs[i].radius < 0 || specs[i].radius > 3)
            return fail(WMD_ERR_BAD_ARG, "wmd_mask_dilate_multi: spec %d (up=%d radius=%d)", i, specs[i].up, specs[i].radius);
        a.s[i] = specs[i];
        maxpix = std::max(maxpix, h * specs[i].up * w * specs[i].up);
    }
    ProfScope prof("mask_dilate_multi_kernel", 25.0 * maxpix * n, 2.0 * maxpix * n, (hipStream_t)stream);
    hipLaunchKernelGGL(mask_dilate_multi_kernel, dim3(std::min((maxpix + 255) / 256, 1024), n), dim3(256), 0,
                       (hipStream_t)stream, mask, h, w, a);
    return check_launch("mask_dilate_multi_kernel");
}

extern "C" int wmd_mask_compact_multi(const wmd_compact_spec* specs, int n, void* stream) {
    if (!specs) return fail(WMD_ERR_BAD_ARG, "wmd_mask_compact_multi: null pointer");
    if (n <= 0 || n > 8) return fail(WMD_ERR_BAD_SHAPE, "wmd_mask_compact_multi: n=%d", n);
    CompactKArgs a;
    double px = 0;
    for (int i = 0; i < n; ++i) {
        if (!specs[i].mask || !specs[i].coords || !specs[i].nnz || specs[i].npix <= 0)
            return fail(WMD_ERR_BAD_ARG, "wmd_mask_compact_multi: spec %d", i);
        a.s[i] = specs[i];
        px += specs[i].npix;
    }
    ProfScope prof("mask_compact_multi_kernel", px, 5.0 * px, (hipStream_t)stream);
    hipLaunchKernelGGL(mask_compact_multi_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, a);
    return check_launch("mask_compact_multi_kernel");
}

extern "C" int wmd_sparse_conv(const wmd_sparse_conv_args* g, void* stream) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: null args");
    if (!g->x1 || !g->out_coords || !g->out_nnz || !g->wp || !g->y)
        return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: null pointer");
    if (g->C2 > 0 && !g->x2) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: C2=%d but x2 is null", g->C2);
    if (g->H <= 0 || g->W <= 0 || g->C1 <= 0 || g->C2 < 0 || g->Cout <= 0 || g->max_out < 0)
        return fail(WMD_ERR_BAD_SHAPE, "wmd_sparse_conv: H=%d W=%d C1=%d C2=%d Cout=%d", g->H, g->W, g->C1, g->C2, g->Cout);
    if (g->ksize != 1 && g->ksize != 3) return fail(WMD_ERR_UNSUPPORTED, "wmd_sparse_conv: ksize=%d", g->ksize);
    if (g->up1 != 1 && g->up1 != 2) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: up1=%d", g->up1);
    if (g->c1_off < 0 || g->c1_off + g->C1 > g->C1tot) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: channel slice");
    if (g->wp2 && (g->Cout > 16 || g->c1_off2 < 0 || g->c1_off2 + g->C1 > g->C1tot || g->C2 != 0))
        return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: dual-head mode needs Cout <= 16, C2 == 0 and a valid slice");
    if (g->pad_mode < 0 || g->pad_mode > 2 || g->act < 0 || g->act > 3) return fail(WMD_ERR_BAD_ARG, "wmd_sparse_conv: enum");
    if (g->max_out == 0) return WMD_OK;
    SparseKArgs a;
    a.g = *g;
    const int Cin = g->C1 + g->C2;
    a.nci4 = ((Cin + 15) / 16) * 4;
    a.ncot = (g->Cout + 15) / 16;
    a.W1 = g->W / g->up1;
    a.plane = (size_t)g->H * g->W;
    a.plane1 = (size_t)(g->H / g->up1) * a.W1;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (g->max_out + 15) / 16;
    const int taps = g->ksize == 3 ? 9 : 1;
    ProfScope prof("sparse_conv_kernel", 0.0, 0.0, s);
    constexpr int WK = 8;  // waves per task for 3x3 (Cin*9 gathers per pixel); 1x1 chains are 9x shorter
    if (g->wp2) {
        if (taps == 9) hipLaunchKernelGGL((sparse_conv_kernel<1, 9, true, WK>), dim3(tiles, 1), dim3(64 * WK), 0, s, a);
        else hipLaunchKernelGGL((sparse_conv_kernel<1, 1, true, 2>), dim3(tiles, 1), dim3(128), 0, s, a);
    } else if (a.ncot >= 4) {
        const dim3 grid(tiles, (a.ncot + 3) / 4);
        if (taps == 9) hipLaunchKernelGGL((sparse_conv_kernel<4, 9, false, WK>), grid, dim3(64 * WK), 0, s, a);
        else hipLaunchKernelGGL((sparse_conv_kernel<4, 1, false, 2>), grid, dim3(128), 0, s, a);
    } else if (a.ncot >= 2) {
        const dim3 grid(tiles, (a.ncot + 1) / 2);
        if (taps == 9) hipLaunchKernelGGL((sparse_conv_kernel<2, 9, false, WK>), grid, dim3(64 * WK), 0, s, a);
        else hipLaunchKernelGGL((sparse_conv_kernel<2, 1, false, 2>), grid, dim3(128), 0, s, a);
    } else {
        const dim3 grid(tiles, 1);
        if (taps == 9) hipLaunchKernelGGL((sparse_conv_kernel<1, 9, false, WK>), grid, dim3(64 * WK), 0, s, a);
        else hipLaunchKernelGGL((sparse_conv_kernel<1, 1, false, 2>), grid, dim3(128), 0, s, a);
    }
    return check_launch("sparse_conv_kernel");
}
