// Photometric loss stack of the KITTI trainer -- what the training step spends its GPU time on
// once the decoder is fast; all HBM-bound gather / stencil kernels with hand-written backward passes:
//   ssim_*        SSIM (KITTI/layers.py:281-311) and compute_reprojection_loss (KITTI/trainer.py:393-405):
//                 0.85 * mean_c clamp((1 - SSIM)/2, 0, 1) + 0.15 * mean_c |target - pred|, in one pass over the two images
//   warp_*        BackprojectDepth -> Project3D -> F.grid_sample(padding_mode="border") (layers.py:176-229,
//                 trainer.py:352-372) fused: depth map + intrinsics + pose -> warped source frame, no point cloud and no
//                 sampling grid in HBM; backward to the depth map and to the 4x4 pose
//   smooth_*      get_smooth_loss (layers.py:238-252): edge-aware first-order smoothness of the mean-normalised disparity
#include <algorithm>
#include <cmath>
#include <cstdint>
#include "wmd_internal.h"

namespace wmd {

__device__ __forceinline__ int refl1(int g, int n) {   // ReflectionPad2d(1) source index of padded coordinate g in [-1, n]
    return g < 0 ? -g : (g >= n ? 2 * n - 2 - g : g);
}

// ------------------------------------------------------------------------------------------------
// SSIM / reprojection loss
// ------------------------------------------------------------------------------------------------
struct SsimStats {
    float mx, my, ex2, ey2, exy;
};

__device__ __forceinline__ SsimStats ssim_window(const float* __restrict__ xp, const float* __restrict__ yp, int y, int x, int H, int W) {
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int ry = refl1(y + dy, H) * W;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int rx = refl1(x + dx, W);
            const float a = xp[ry + rx], b = yp[ry + rx];
            sx += a;
            sy += b;
            sxx += a * a;
            syy += b * b;
            sxy += a * b;
        }
    }
    const float inv9 = 1.f / 9.f;
    return SsimStats{sx * inv9, sy * inv9, sxx * inv9, syy * inv9, sxy * inv9};
}

constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

__device__ __forceinline__ float ssim_value(const SsimStats& s, float* n1o = nullptr, float* n2o = nullptr, float* d1o = nullptr,
                                            float* d2o = nullptr) {
    const float sig_x = s.ex2 - s.mx * s.mx, sig_y = s.ey2 - s.my * s.my, sig_xy = s.exy - s.mx * s.my;
    const float n1 = 2.f * s.mx * s.my + kC1, n2 = 2.f * sig_xy + kC2;
    const float d1 = s.mx * s.mx + s.my * s.my + kC1, d2 = sig_x + sig_y + kC2;
    if (n1o) *n1o = n1, *n2o = n2, *d1o = d1, *d2o = d2;
    return (n1 * n2) / (d1 * d2);
}

// mode 0: out [B,C,H,W] = clamp((1 - SSIM)/2, 0, 1)          (the SSIM module)
// mode 1: out [B,1,H,W] = w_ssim * mean_c(...) + w_l1 * mean_c |y - x|   (compute_reprojection_loss)
__global__ void ssim_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int B, int C,
                                int H, int W, int mode, float w_ssim, float w_l1) {
    const int plane = H * W;
    const int total = B * plane * (mode == 0 ? C : 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int p = i % plane, yy = p / W, xx = p - yy * W;
        if (mode == 0) {
            const size_t base = (size_t)(i / plane) * plane;   // (b, c) plane
            const float s = ssim_value(ssim_window(x + base, y + base, yy, xx, H, W));
            out[i] = fminf(fmaxf((1.f - s) * 0.5f, 0.f), 1.f);
        } else {
            const int b = i / plane;
            float ss = 0.f, l1 = 0.f;
            for (int c = 0; c < C; ++c) {
                const size_t base = ((size_t)b * C + c) * plane;
                const float s = ssim_value(ssim_window(x + base, y + base, yy, xx, H, W));
                ss += fminf(fmaxf((1.f - s) * 0.5f, 0.f), 1.f);
                l1 += fabsf(y[base + p] - x[base + p]);
            }
            out[i] = w_ssim * (ss / C) + w_l1 * (l1 / C);
        }
    }
}

// backward stage 1: per window q, the upstream-weighted partials of the loss w.r.t. the window statistics of x:
//   coef[0] = g * dL/dmu_x,  coef[1] = g * dL/dE[x^2],  coef[2] = g * dL/dE[xy]         (planes of [B,C,H,W])
__global__ void ssim_bwd_coef_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g,
                                     float* __restrict__ coef, int B, int C, int H, int W, int mode, float w_ssim) {
    const int plane = H * W;
    const size_t n = (size_t)B * C * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % plane), yy = p / W, xx = p - yy * W;
        const size_t base = i - p;
        const int b = (int)(i / ((size_t)C * plane));
        const SsimStats s = ssim_window(x + base, y + base, yy, xx, H, W);
        float n1, n2, d1, d2;
        const float S = ssim_value(s, &n1, &n2, &d1, &d2);
        const float L = (1.f - S) * 0.5f;
        const float up = mode == 0 ? g[i] : g[(size_t)b * plane + p] * (w_ssim / C);
        const float dLdS = (L >= 0.f && L <= 1.f) ? -0.5f * up : 0.f;   // clamp passes the gradient on [0, 1]
        const float inv = 1.f / (d1 * d2);
        const float dS_dmx = (2.f * s.my * n2 - 2.f * s.my * n1) * inv - S * (2.f * s.mx / d1 - 2.f * s.mx / d2);
        const float dS_dex2 = -S / d2;
        const float dS_dexy = 2.f * n1 * inv;
        coef[i] = dLdS * dS_dmx;
        coef[n + i] = dLdS * dS_dex2;
        coef[2 * n + i] = dLdS * dS_dexy;
    }
}

// backward stage 2: dx[p] = (1/9) * sum over the windows q that contain p (with the multiplicity reflection padding gives
// border pixels) of (coef0[q] + 2 x[p] coef1[q] + y[p] coef2[q])  +  the L1 term of mode 1.
__global__ void ssim_bwd_gather_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g,
                                       const float* __restrict__ coef, float* __restrict__ dx, int B, int C, int H, int W,
                                       int mode, float w_l1) {
    const int plane = H * W;
    const size_t n = (size_t)B * C * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % plane), py = p / W, px = p - py * W;
        const size_t base = i - p;
        float sa = 0.f, sb = 0.f, sc = 0.f;
        for (int qy = max(py - 1, 0); qy <= min(py + 1, H - 1); ++qy) {   // a reflected tap lands at most one line away
            int my = 0;   // how many of the three rows of window qy are (reflections of) row py
            for (int t = -1; t <= 1; ++t) my += refl1(qy + t, H) == py ? 1 : 0;
            if (!my) continue;
            for (int qx = max(px - 1, 0); qx <= min(px + 1, W - 1); ++qx) {
                int mx = 0;
                for (int t = -1; t <= 1; ++t) mx += refl1(qx + t, W) == px ? 1 : 0;
                if (!mx) continue;
                const float m = (float)(my * mx);
                const size_t q = base + (size_t)qy * W + qx;
                sa += m * coef[q];
                sb += m * coef[n + q];
                sc += m * coef[2 * n + q];
            }
        }
        const float xv = x[i], yv = y[i];
        float d = (sa + 2.f * xv * sb + yv * sc) * (1.f / 9.f);
        if (mode == 1) {
            const int b = (int)(i / ((size_t)C * plane));
            const float diff = yv - xv;   // |other - this|: d/d(this) = -sign(other - this), whichever operand `x` is here
            d -= g[(size_t)b * plane + p] * (w_l1 / C) * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
        }
        dx[i] = d;
    }
}

}  // namespace wmd

using namespace wmd;

static int ssim_check(const char* who, const void* x, const void* y, int B, int C, int H, int W, int mode) {
    if (!x || !y) return fail(WMD_ERR_BAD_ARG, "%s: null tensor pointer", who);
    if (B <= 0 || C <= 0 || H < 2 || W < 2) return fail(WMD_ERR_BAD_SHAPE, "%s: B=%d C=%d H=%d W=%d (reflection padding needs H,W >= 2)", who, B, C, H, W);
    if (mode != 0 && mode != 1) return fail(WMD_ERR_BAD_ARG, "%s: mode=%d", who, mode);
    if ((double)B * C * H * W > 2147483647.0) return fail(WMD_ERR_UNSUPPORTED, "%s: more than 2^31 elements", who);
    return WMD_OK;
}

extern "C" int wmd_ssim_fwd(const float* x, const float* y, float* out, int B, int C, int H, int W, int mode, float w_ssim,
                            float w_l1, void* stream) {
    int st = ssim_check("wmd_ssim_fwd", x, y, B, C, H, W, mode);
    if (st) return st;
    if (!out) return fail(WMD_ERR_BAD_ARG, "wmd_ssim_fwd: null output");
    const size_t n = (size_t)B * H * W * (mode == 0 ? C : 1);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("ssim_fwd_kernel", 60.0 * B * C * H * W, 4.0 * (2.0 * B * C * H * W + n), s);
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, x, y, out, B, C, H, W, mode,
                       w_ssim, w_l1);
    return check_launch("ssim_fwd_kernel");
}

extern "C" size_t wmd_ssim_bwd_workspace_floats(int B, int C, int H, int W) {
    return (B > 0 && C > 0 && H > 0 && W > 0) ? (size_t)3 * B * C * H * W : 0;
}

extern "C" int wmd_ssim_bwd(const float* x, const float* y, const float* g, float* dx, float* dy, int B, int C, int H, int W, int mode,
                            float w_ssim, float w_l1, float* workspace, size_t workspace_floats, void* stream) {
    int st = ssim_check("wmd_ssim_bwd", x, y, B, C, H, W, mode);
    if (st) return st;
    if (!g || (!dx && !dy)) return fail(WMD_ERR_BAD_ARG, "wmd_ssim_bwd: null gradient pointer");
    const size_t n = (size_t)B * C * H * W;
    if (!workspace || workspace_floats < 3 * n) return fail(WMD_ERR_WORKSPACE, "wmd_ssim_bwd: workspace %zu < %zu floats", workspace_floats, 3 * n);
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
    // SSIM is symmetric in its two images: the gradient w.r.t. y is the same computation with the operands swapped
    for (int side = 0; side < 2; ++side) {
        float* d = side == 0 ? dx : dy;
        if (!d) continue;
        const float* a = side == 0 ? x : y;
        const float* b = side == 0 ? y : x;
        {
            ProfScope prof("ssim_bwd_coef_kernel", 90.0 * n, 4.0 * 6.0 * n, s);
            hipLaunchKernelGGL(ssim_bwd_coef_kernel, dim3(blocks), dim3(256), 0, s, a, b, g, workspace, B, C, H, W, mode, w_ssim);
        }
        st = check_launch("ssim_bwd_coef_kernel");
        if (st) return st;
        ProfScope prof("ssim_bwd_gather_kernel", 100.0 * n, 4.0 * 7.0 * n, s);
        hipLaunchKernelGGL(ssim_bwd_gather_kernel, dim3(blocks), dim3(256), 0, s, a, b, g, workspace, d, B, C, H, W, mode, w_l1);
        st = check_launch("ssim_bwd_gather_kernel");
        if (st) return st;
    }
    return WMD_OK;
}

// ------------------------------------------------------------------------------------------------
// warp: BackprojectDepth -> Project3D -> grid_sample(bilinear, padding_mode="border", align_corners=False)
// ------------------------------------------------------------------------------------------------
namespace wmd {

struct WarpGeom {       // everything about one target pixel that forward and backward share
    float rx, ry, rz;   // inv_K[:3,:3] * (x, y, 1)
    float X, Y, Z;      // depth * r
    float u, v, w;      // P * (X, Y, Z, 1)
    float ix, iy;       // clipped source coordinates
    float mx, my;       // clip gradient multipliers (0 where the border clamp is active)
};

__device__ __forceinline__ void warp_P(const float* __restrict__ K, const float* __restrict__ T, float* P) {   // (K T)[:3,:]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += K[i * 4 + k] * T[k * 4 + j];
            P[i * 4 + j] = s;
        }
}

__device__ __forceinline__ float clip_coord(float in, int size, float* mult) {   // ATen clip_coordinates_set_grad
    // written so that a NaN coordinate (depth 0 x inf, a degenerate pose) lands on 0 with zero gradient instead of
    // slipping through both comparisons and being cast to an int: ATen never indexes with a NaN either
    const float mx = (float)(size - 1);
    const bool inside = in > 0.f && in < mx;          // false for NaN
    *mult = inside ? 1.f : 0.f;
    return inside ? in : (in >= mx ? mx : 0.f);       // NaN -> 0
}

__device__ __forceinline__ WarpGeom warp_geom(float depth, int x, int y, const float* __restrict__ iK, const float* P, int H, int W,
                                              int Hs, int Ws, float eps) {
    WarpGeom g;
    g.rx = iK[0] * x + iK[1] * y + iK[2];
    g.ry = iK[4] * x + iK[5] * y + iK[6];
    g.rz = iK[8] * x + iK[9] * y + iK[10];
    g.X = depth * g.rx;
    g.Y = depth * g.ry;
    g.Z = depth * g.rz;
    g.u = P[0] * g.X + P[1] * g.Y + P[2] * g.Z + P[3];
    g.v = P[4] * g.X + P[5] * g.Y + P[6] * g.Z + P[7];
    g.w = P[8] * g.X + P[9] * g.Y + P[10] * g.Z + P[11];
    const float den = g.w + eps;
    float gx = (g.u / den) / (float)(W - 1), gy = (g.v / den) / (float)(H - 1);   // Project3D: /= (width - 1), (x - 0.5) * 2
    gx = (gx - 0.5f) * 2.f;
    gy = (gy - 0.5f) * 2.f;
    const float ux = ((gx + 1.f) * Ws - 1.f) * 0.5f, uy = ((gy + 1.f) * Hs - 1.f) * 0.5f;   // grid_sampler_unnormalize
    g.ix = clip_coord(ux, Ws, &g.mx);
    g.iy = clip_coord(uy, Hs, &g.my);
    return g;
}

__global__ void warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ depth, const float* __restrict__ K,
                                const float* __restrict__ iK, const float* __restrict__ T, float* __restrict__ out, int B, int C,
                                int H, int W, int Hs, int Ws, float eps) {
    const int b = blockIdx.y;
    float P[12];
    warp_P(K + b * 16, T + b * 16, P);
    const int plane = H * W, splane = Hs * Ws;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += gridDim.x * blockDim.x) {
        const int y = p / W, x = p - y * W;
        const WarpGeom g = warp_geom(depth[(size_t)b * plane + p], x, y, iK + b * 16, P, H, W, Hs, Ws, eps);
        const float fx0 = floorf(g.ix), fy0 = floorf(g.iy);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float tx = g.ix - fx0, ty = g.iy - fy0;
        const bool in_x1 = x1 < Ws, in_y1 = y1 < Hs;   // x0, y0 are inside after the clip
        for (int c = 0; c < C; ++c) {
            const float* s = src + ((size_t)b * C + c) * splane;
            const float v00 = s[y0 * Ws + x0], v01 = in_x1 ? s[y0 * Ws + x1] : 0.f;
            const float v10 = in_y1 ? s[y1 * Ws + x0] : 0.f, v11 = (in_x1 && in_y1) ? s[y1 * Ws + x1] : 0.f;
            out[((size_t)b * C + c) * plane + p] = v00 * (1.f - tx) * (1.f - ty) + v01 * tx * (1.f - ty) + v10 * (1.f - tx) * ty + v11 * tx * ty;
        }
    }
}

// d_depth per pixel; dP (3x4 per image) as per-block partial sums -> warp_bwd_finish_kernel
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                       const float* __restrict__ K, const float* __restrict__ iK,
                                                       const float* __restrict__ T, const float* __restrict__ gout,
                                                       float* __restrict__ ddepth, float* __restrict__ partial, int B, int C, int H,
                                                       int W, int Hs, int Ws, float eps) {
    const int b = blockIdx.y;
    float P[12];
    warp_P(K + b * 16, T + b * 16, P);
    const int plane = H * W, splane = Hs * Ws;
    float dP[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) dP[k] = 0.f;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += gridDim.x * blockDim.x) {
        const int y = p / W, x = p - y * W;
        const float dep = depth[(size_t)b * plane + p];
        const WarpGeom g = warp_geom(dep, x, y, iK + b * 16, P, H, W, Hs, Ws, eps);
        const float fx0 = floorf(g.ix), fy0 = floorf(g.iy);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float tx = g.ix - fx0, ty = g.iy - fy0;
        const bool in_x1 = x1 < Ws, in_y1 = y1 < Hs;
        float gix = 0.f, giy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* s = src + ((size_t)b * C + c) * splane;
            const float v00 = s[y0 * Ws + x0], v01 = in_x1 ? s[y0 * Ws + x1] : 0.f;
            const float v10 = in_y1 ? s[y1 * Ws + x0] : 0.f, v11 = (in_x1 && in_y1) ? s[y1 * Ws + x1] : 0.f;
            const float go = gout[((size_t)b * C + c) * plane + p];
            gix += go * ((v01 - v00) * (1.f - ty) + (v11 - v10) * ty);
            giy += go * ((v10 - v00) * (1.f - tx) + (v11 - v01) * tx);
        }
        // clip -> unnormalize (size / 2) -> (g - 0.5) * 2 and / (size_target - 1)
        const float dgx = gix * g.mx * (0.5f * Ws) * (2.f / (float)(W - 1));
        const float dgy = giy * g.my * (0.5f * Hs) * (2.f / (float)(H - 1));
        const float den = g.w + eps, inv = 1.f / den;
        const float du = dgx * inv, dv = dgy * inv, dw = -(dgx * g.u + dgy * g.v) * inv * inv;
        const float dX = P[0] * du + P[4] * dv + P[8] * dw;
        const float dY = P[1] * du + P[5] * dv + P[9] * dw;
        const float dZ = P[2] * du + P[6] * dv + P[10] * dw;
        ddepth[(size_t)b * plane + p] = dX * g.rx + dY * g.ry + dZ * g.rz;
        dP[0] += du * g.X, dP[1] += du * g.Y, dP[2] += du * g.Z, dP[3] += du;
        dP[4] += dv * g.X, dP[5] += dv * g.Y, dP[6] += dv * g.Z, dP[7] += dv;
        dP[8] += dw * g.X, dP[9] += dw * g.Y, dP[10] += dw * g.Z, dP[11] += dw;
    }
    __shared__ float red[4][12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        float s = dP[k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12)
        partial[((size_t)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// dT[k][j] = sum_{i<3} K[i][k] * dP[i][j]      (P = (K T)[:3,:]), dP summed over the blocks in fixed order
__global__ __launch_bounds__(64) void warp_bwd_finish_kernel(const float* __restrict__ partial, const float* __restrict__ K,
                                                             float* __restrict__ dT, int B, int nblk) {
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ float dP[12];
    for (int e = 0; e < 12; ++e) {   // lane-strided partial sums in a fixed order, then a shuffle tree (deterministic)
        float s = 0.f;
        for (int j = lane; j < nblk; j += 64) s += partial[((size_t)b * nblk + j) * 12 + e];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) dP[e] = s;
    }
    __syncthreads();
    if (lane < 16) {
        const int k = lane >> 2, j = lane & 3;
        float s = 0.f;
        for (int i = 0; i < 3; ++i) s += K[b * 16 + i * 4 + k] * dP[i * 4 + j];
        dT[b * 16 + lane] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// edge-aware smoothness
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float img_grad(const float* __restrict__ img, size_t base, int plane, int C, int p, int q) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += fabsf(img[base + (size_t)c * plane + p] - img[base + (size_t)c * plane + q]);
    return s / C;
}

// partial[blk][0] = sum |d(p) - d(p + x)| exp(-gamma gx),  partial[blk][1] = same along y
__global__ __launch_bounds__(256) void smooth_fwd_kernel(const float* __restrict__ disp, const float* __restrict__ img,
                                                         double* __restrict__ partial, int B, int C, int H, int W, float gamma) {
    const int plane = H * W;
    const int total = B * plane;
    double sx = 0.0, sy = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / plane, p = i - b * plane, y = p / W, x = p - y * W;
        const size_t ib = (size_t)b * C * plane;
        const float d = disp[i];
        if (x + 1 < W) sx += (double)(fabsf(d - disp[i + 1]) * expf(-gamma * img_grad(img, ib, plane, C, p, p + 1)));
        if (y + 1 < H) sy += (double)(fabsf(d - disp[i + W]) * expf(-gamma * img_grad(img, ib, plane, C, p, p + W)));
    }
    __shared__ double red[4][2];
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][0] = sx, red[threadIdx.x >> 6][1] = sy;
    __syncthreads();
    if (threadIdx.x < 2) partial[blockIdx.x * 2 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void smooth_finish_kernel(const double* __restrict__ partial, float* __restrict__ out, int nblk,
                                                           double nx, double ny) {
    // one wavefront: lane-strided sums in a fixed order, then a shuffle tree (deterministic)
    double sx = 0.0, sy = 0.0;
    for (int j = threadIdx.x; j < nblk; j += 64) {
        sx += partial[j * 2];
        sy += partial[j * 2 + 1];
    }
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
    }
    if (threadIdx.x == 0) out[0] = (float)(sx / nx + sy / ny);
}

// ddisp[p] = g * ( [sign(d_p - d_{p+x}) w_x(p) - sign(d_{p-x} - d_p) w_x(p-x)] / Nx + the same along y )
__global__ void smooth_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ img, const float* __restrict__ gout,
                                  float* __restrict__ ddisp, int B, int C, int H, int W, float gamma) {
    const int plane = H * W;
    const int total = B * plane;
    const float g = gout[0];
    const float inx = 1.f / ((float)B * H * (W - 1)), iny = 1.f / ((float)B * (H - 1) * W);
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / plane, p = i - b * plane, y = p / W, x = p - y * W;
        const size_t ib = (size_t)b * C * plane;
        const float d = disp[i];
        float acc = 0.f;
        if (x + 1 < W) acc += sgn(d - disp[i + 1]) * expf(-gamma * img_grad(img, ib, plane, C, p, p + 1)) * inx;
        if (x > 0) acc -= sgn(disp[i - 1] - d) * expf(-gamma * img_grad(img, ib, plane, C, p - 1, p)) * inx;
        if (y + 1 < H) acc += sgn(d - disp[i + W]) * expf(-gamma * img_grad(img, ib, plane, C, p, p + W)) * iny;
        if (y > 0) acc -= sgn(disp[i - W] - d) * expf(-gamma * img_grad(img, ib, plane, C, p - W, p)) * iny;
        ddisp[i] = g * acc;
    }
}

}  // namespace wmd

static int warp_check(const char* who, const wmd_warp_args* g) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "%s: null args", who);
    if (!g->src || !g->depth || !g->K || !g->inv_K || !g->T) return fail(WMD_ERR_BAD_ARG, "%s: null tensor pointer", who);
    if (g->B <= 0 || g->C <= 0 || g->H < 2 || g->W < 2 || g->Hs <= 0 || g->Ws <= 0)
        return fail(WMD_ERR_BAD_SHAPE, "%s: B=%d C=%d H=%d W=%d Hs=%d Ws=%d", who, g->B, g->C, g->H, g->W, g->Hs, g->Ws);
    if ((double)g->B * g->C * std::max((double)g->H * g->W, (double)g->Hs * g->Ws) > 2147483647.0)
        return fail(WMD_ERR_UNSUPPORTED, "%s: more than 2^31 elements", who);
    return WMD_OK;
}

static int warp_blocks(int plane) { return std::max(1, std::min((plane + 1023) / 1024, 256)); }

extern "C" int wmd_warp_fwd(const wmd_warp_args* g, float* out, void* stream) {
    int st = warp_check("wmd_warp_fwd", g);
    if (st) return st;
    if (!out) return fail(WMD_ERR_BAD_ARG, "wmd_warp_fwd: null output");
    hipStream_t s = (hipStream_t)stream;
    const double n = (double)g->B * g->H * g->W;
    ProfScope prof("warp_fwd_kernel", n * (60.0 + 8.0 * g->C), 4.0 * n * (1.0 + 2.0 * g->C), s);
    hipLaunchKernelGGL(warp_fwd_kernel, dim3(warp_blocks(g->H * g->W), g->B), dim3(256), 0, s, g->src, g->depth, g->K, g->inv_K, g->T, out,
                       g->B, g->C, g->H, g->W, g->Hs, g->Ws, g->eps);
    return check_launch("warp_fwd_kernel");
}

extern "C" size_t wmd_warp_bwd_workspace_floats(const wmd_warp_args* g) {
    return g ? (size_t)g->B * warp_blocks(g->H * g->W) * 12 : 0;
}

extern "C" int wmd_warp_bwd(const wmd_warp_args* g, const float* grad_out, float* ddepth, float* dT, float* workspace,
                            size_t workspace_floats, void* stream) {
    int st = warp_check("wmd_warp_bwd", g);
    if (st) return st;
    if (!grad_out || !ddepth || !dT) return fail(WMD_ERR_BAD_ARG, "wmd_warp_bwd: null gradient pointer");
    const int nblk = warp_blocks(g->H * g->W);
    if (!workspace || workspace_floats < (size_t)g->B * nblk * 12)
        return fail(WMD_ERR_WORKSPACE, "wmd_warp_bwd: workspace %zu < %zu floats", workspace_floats, (size_t)g->B * nblk * 12);
    hipStream_t s = (hipStream_t)stream;
    const double n = (double)g->B * g->H * g->W;
    {
        ProfScope prof("warp_bwd_kernel", n * (120.0 + 14.0 * g->C), 4.0 * n * (2.0 + 2.0 * g->C), s);
        hipLaunchKernelGGL(warp_bwd_kernel, dim3(nblk, g->B), dim3(256), 0, s, g->src, g->depth, g->K, g->inv_K, g->T, grad_out, ddepth,
                           workspace, g->B, g->C, g->H, g->W, g->Hs, g->Ws, g->eps);
    }
    st = check_launch("warp_bwd_kernel");
    if (st) return st;
    hipLaunchKernelGGL(warp_bwd_finish_kernel, dim3(g->B), dim3(64), 0, s, workspace, g->K, dT, g->B, nblk);
    return check_launch("warp_bwd_finish_kernel");
}

static int smooth_blocks(int total) { return std::max(1, std::min((total + 2047) / 2048, 512)); }

extern "C" size_t wmd_smooth_workspace_floats(int B, int H, int W) {
    return (B > 0 && H > 0 && W > 0) ? (size_t)smooth_blocks(B * H * W) * 2 * 2 + 2 : 0;   // fp64 partials (+ alignment slack)
}

extern "C" int wmd_smooth_fwd(const float* disp, const float* img, float* out, int B, int C, int H, int W, float gamma, float* workspace,
                              size_t workspace_floats, void* stream) {
    if (!disp || !img || !out) return fail(WMD_ERR_BAD_ARG, "wmd_smooth_fwd: null tensor pointer");
    if (B <= 0 || C <= 0 || H < 2 || W < 2) return fail(WMD_ERR_BAD_SHAPE, "wmd_smooth_fwd: B=%d C=%d H=%d W=%d", B, C, H, W);
    if ((double)B * C * H * W > 2147483647.0) return fail(WMD_ERR_UNSUPPORTED, "wmd_smooth_fwd: more than 2^31 elements");
    if (!workspace || workspace_floats < wmd_smooth_workspace_floats(B, H, W))
        return fail(WMD_ERR_WORKSPACE, "wmd_smooth_fwd: workspace %zu < %zu floats", workspace_floats, wmd_smooth_workspace_floats(B, H, W));
    hipStream_t s = (hipStream_t)stream;
    const int nblk = smooth_blocks(B * H * W);
    double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 7) & ~(uintptr_t)7);
    const double n = (double)B * H * W;
    ProfScope prof("smooth_fwd_kernel", n * (8.0 + 6.0 * C), 4.0 * n * (1.0 + C), s);
    hipLaunchKernelGGL(smooth_fwd_kernel, dim3(nblk), dim3(256), 0, s, disp, img, partial, B, C, H, W, gamma);
    hipLaunchKernelGGL(smooth_finish_kernel, dim3(1), dim3(64), 0, s, partial, out, nblk, (double)B * H * (W - 1), (double)B * (H - 1) * W);
    return check_launch("smooth_fwd_kernel");
}

extern "C" int wmd_smooth_bwd(const float* disp, const float* img, const float* grad_out, float* ddisp, int B, int C, int H, int W,
                              float gamma, void* stream) {
    if (!disp || !img || !grad_out || !ddisp) return fail(WMD_ERR_BAD_ARG, "wmd_smooth_bwd: null tensor pointer");
    if (B <= 0 || C <= 0 || H < 2 || W < 2) return fail(WMD_ERR_BAD_SHAPE, "wmd_smooth_bwd: B=%d C=%d H=%d W=%d", B, C, H, W);
    if ((double)B * C * H * W > 2147483647.0) return fail(WMD_ERR_UNSUPPORTED, "wmd_smooth_bwd: more than 2^31 elements");
    hipStream_t s = (hipStream_t)stream;
    const double n = (double)B * H * W;
    ProfScope prof("smooth_bwd_kernel", n * (16.0 + 12.0 * C), 4.0 * n * (2.0 + C), s);
    hipLaunchKernelGGL(smooth_bwd_kernel, dim3(smooth_blocks(B * H * W)), dim3(256), 0, s, disp, img, grad_out, ddisp, B, C, H, W, gamma);
    return check_launch("smooth_bwd_kernel");
}
