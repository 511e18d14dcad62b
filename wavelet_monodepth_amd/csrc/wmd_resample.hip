// Multi-scale loss front-end: bilinear up-sampling of a decoder output to the loss
// resolution, optionally fused with disp_to_depth.
//   KITTI  trainer.py:333-344  F.interpolate(disp, [H, W], mode="bilinear", align_corners=False) then
//          layers.py:16-25     depth = 1 / (min_disp + (max_disp - min_disp) * disp)
//   NYUv2  train.py:304-306    F.interpolate(disp, scale_factor=2**s, mode="bilinear", align_corners=True)
// Index arithmetic follows ATen's upsample_bilinear2d (area_pixel_compute_source_index): HBM-bound, one thread
// per output pixel forward; the backward is the exact adjoint in gather form (one thread per INPUT pixel walks
// the output window whose footprint touches it — no atomics, deterministic).
#include <algorithm>
#include "wmd_internal.h"

namespace wmd {

__device__ __forceinline__ float src_index(float scale, int dst, int align_corners) {
    if (align_corners) return scale * dst;
    const float s = scale * (dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

__global__ void upsample_bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ depth,
                                             int N, int h, int w, int H, int W, float sh, float sw, int align_corners,
                                             float min_disp, float max_disp) {
    const size_t total = (size_t)N * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = i % W, oy = (i / W) % H;
        const size_t n = i / ((size_t)H * W);
        const float fy = src_index(sh, oy, align_corners), fx = src_index(sw, ox, align_corners);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0;
        const float* p = x + n * h * w;
        const float v = (1.f - ly) * ((1.f - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1]) +
                        ly * ((1.f - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1]);
        if (y) y[i] = v;
        if (depth) depth[i] = 1.f / (min_disp + (max_disp - min_disp) * v);
    }
}

__global__ void upsample_bilinear_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ddepth,
                                             const float* __restrict__ depth, float* __restrict__ dx, int N, int h, int w,
                                             int H, int W, float sh, float sw, int align_corners, float min_disp,
                                             float max_disp) {
    const size_t total = (size_t)N * h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ix = i % w, iy = (i / w) % h;
        const size_t n = i / ((size_t)h * w);
        // output rows/cols whose source coordinate lies in (i-1, i+1): invert src = s*(o+0.5)-0.5 (or s*o), +-1 margin;
        // the weights computed below decide exactly
        const float off = align_corners ? 0.f : 0.5f;
        int oy_lo = 0, oy_hi = H - 1, ox_lo = 0, ox_hi = W - 1;
        if (sh > 0.f) {
            oy_lo = max(0, (int)floorf((iy - 1 + off) / sh - off) - 1);
            oy_hi = min(H - 1, (int)ceilf((iy + 1 + off) / sh - off) + 1);
        }
        if (sw > 0.f) {
            ox_lo = max(0, (int)floorf((ix - 1 + off) / sw - off) - 1);
            ox_hi = min(W - 1, (int)ceilf((ix + 1 + off) / sw - off) + 1);
        }
        if (iy == 0) oy_lo = 0;          // sources clamped to 0 all land on the first row / column
        if (ix == 0) ox_lo = 0;
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float fy = src_index(sh, oy, align_corners);
            const int y0 = (int)fy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const float ly = fy - y0;
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const float fx = src_index(sw, ox, align_corners);
                const int x0 = (int)fx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float lx = fx - x0;
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                if (wx == 0.f) continue;
                const size_t o = (n * H + oy) * W + ox;
                float g = dy ? dy[o] : 0.f;
                if (ddepth) {
                    const float d = depth[o];
                    g -= ddepth[o] * (max_disp - min_disp) * d * d;  // d(1/s)/ddisp = -(max-min)/s^2 = -(max-min)*depth^2
                }
                acc += wy * wx * g;
            }
        }
        dx[i] = acc;
    }
}

static float area_scale(int in, int out, int align_corners) {
    if (align_corners) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}

}  // namespace wmd

using namespace wmd;

extern "C" int wmd_upsample_bilinear_fwd(const float* x, float* y, float* depth, int N, int h, int w, int H, int W,
                                         int align_corners, float min_depth, float max_depth, void* stream) {
    if (!x || (!y && !depth)) return fail(WMD_ERR_BAD_ARG, "wmd_upsample_bilinear_fwd: null pointer");
    if (N < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_upsample_bilinear_fwd: sizes");
    if (depth && !(min_depth > 0.f && max_depth > min_depth)) return fail(WMD_ERR_BAD_ARG, "wmd_upsample_bilinear_fwd: depth range");
    if (N == 0) return WMD_OK;
    const size_t n = (size_t)N * H * W;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("upsample_bilinear_fwd_kernel", 8.0 * n, 4.0 * (n * ((y ? 1 : 0) + (depth ? 1 : 0)) + (double)N * h * w), s);
    hipLaunchKernelGGL(upsample_bilinear_fwd_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, (size_t)kNumCU * 16)),
                       dim3(256), 0, s, x, y, depth, N, h, w, H, W, area_scale(h, H, align_corners),
                       area_scale(w, W, align_corners), align_corners, depth ? 1.f / max_depth : 0.f, depth ? 1.f / min_depth : 0.f);
    return check_launch("upsample_bilinear_fwd_kernel");
}

extern "C" int wmd_upsample_bilinear_bwd(const float* dy, const float* ddepth, const float* depth, float* dx, int N, int h,
                                         int w, int H, int W, int align_corners, float min_depth, float max_depth,
                                         void* stream) {
    if (!dx || (!dy && !ddepth)) return fail(WMD_ERR_BAD_ARG, "wmd_upsample_bilinear_bwd: null pointer");
    if (ddepth && !depth) return fail(WMD_ERR_BAD_ARG, "wmd_upsample_bilinear_bwd: d(depth) needs the forward depth");
    if (N < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_upsample_bilinear_bwd: sizes");
    if (N == 0) return WMD_OK;
    const size_t n = (size_t)N * h * w;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("upsample_bilinear_bwd_kernel", 8.0 * N * H * W, 4.0 * ((double)N * H * W * (ddepth ? 3 : 1) + n), s);
    hipLaunchKernelGGL(upsample_bilinear_bwd_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, (size_t)kNumCU * 16)),
                       dim3(256), 0, s, dy, ddepth, depth, dx, N, h, w, H, W, area_scale(h, H, align_corners),
                       area_scale(w, W, align_corners), align_corners, ddepth ? 1.f / max_depth : 0.f, ddepth ? 1.f / min_depth : 0.f);
    return check_launch("upsample_bilinear_bwd_kernel");
}
