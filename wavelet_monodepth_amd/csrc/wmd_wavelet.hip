// 2-D Haar synthesis / analysis (one level) for gfx950 — pure HBM streaming kernels.
//
// Arithmetic (pytorch_wavelets DWTInverse/DWTForward with wave="haar"; the authors' closed form is
// KITTI/networks/decoders/depth_decoder.py:225-239): with a=LL, b=LH, c=HL, d=HH
//   y[2i  ,2j] = (a+b+c+d)/2   y[2i  ,2j+1] = (a+b-c-d)/2
//   y[2i+1,2j] = (a-b+c-d)/2   y[2i+1,2j+1] = (a-b-c+d)/2
// The transform is orthonormal, so the adjoint of the synthesis is the analysis butterfly.
//
// Algorithmic traffic: 4 coefficient reads + 4 output writes per 2x2 block = 8 B per output pixel
// (12 B when the normalised `disp` plane is written too).  Each thread handles two horizontally
// adjacent coefficient positions: float2 coefficient loads, float4 output stores, fully coalesced.
#include <algorithm>
#include "wmd_internal.h"

namespace wmd {

__global__ void idwt_haar_fwd_kernel(const float* __restrict__ yl, const float* __restrict__ yh,
                                     float* __restrict__ out, float* __restrict__ disp, int N, int h, int w,
                                     float disp_scale, int clamp01) {
    const int w2 = w >> 1;  // pairs per row (w even) ; odd w handled by the scalar tail kernel
    const size_t total = (size_t)N * h * w2;
    const size_t plane = (size_t)h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int jp = i % w2;
        const size_t r = i / w2;
        const int y = r % h;
        const size_t n = r / h;
        const size_t src = n * plane + (size_t)y * w + 2 * jp;
        const float2 a = *reinterpret_cast<const float2*>(yl + src);
        const float* hb = yh + n * 3 * plane + (size_t)y * w + 2 * jp;
        const float2 b = *reinterpret_cast<const float2*>(hb);
        const float2 c = *reinterpret_cast<const float2*>(hb + plane);
        const float2 d = *reinterpret_cast<const float2*>(hb + 2 * plane);
        float4 top, bot;
        top.x = (a.x + b.x + c.x + d.x) * 0.5f;
        top.y = (a.x + b.x - c.x - d.x) * 0.5f;
        bot.x = (a.x - b.x + c.x - d.x) * 0.5f;
        bot.y = (a.x - b.x - c.x + d.x) * 0.5f;
        top.z = (a.y + b.y + c.y + d.y) * 0.5f;
        top.w = (a.y + b.y - c.y - d.y) * 0.5f;
        bot.z = (a.y - b.y + c.y - d.y) * 0.5f;
        bot.w = (a.y - b.y - c.y + d.y) * 0.5f;
        const size_t dst = n * 4 * plane + (size_t)(2 * y) * (2 * w) + 4 * jp;
        *reinterpret_cast<float4*>(out + dst) = top;
        *reinterpret_cast<float4*>(out + dst + 2 * w) = bot;
        if (disp) {
            float4 t = top, u = bot;
            t.x *= disp_scale; t.y *= disp_scale; t.z *= disp_scale; t.w *= disp_scale;
            u.x *= disp_scale; u.y *= disp_scale; u.z *= disp_scale; u.w *= disp_scale;
            if (clamp01) {
                t.x = fminf(fmaxf(t.x, 0.f), 1.f); t.y = fminf(fmaxf(t.y, 0.f), 1.f);
                t.z = fminf(fmaxf(t.z, 0.f), 1.f); t.w = fminf(fmaxf(t.w, 0.f), 1.f);
                u.x = fminf(fmaxf(u.x, 0.f), 1.f); u.y = fminf(fmaxf(u.y, 0.f), 1.f);
                u.z = fminf(fmaxf(u.z, 0.f), 1.f); u.w = fminf(fmaxf(u.w, 0.f), 1.f);
            }
            *reinterpret_cast<float4*>(disp + dst) = t;
            *reinterpret_cast<float4*>(disp + dst + 2 * w) = u;
        }
    }
}

// scalar form for odd coefficient widths (never hit by the decoders; kept for API completeness)
__global__ void idwt_haar_fwd_scalar_kernel(const float* __restrict__ yl, const float* __restrict__ yh,
                                            float* __restrict__ out, float* __restrict__ disp, int N, int h, int w,
                                            float disp_scale, int clamp01) {
    const size_t total = (size_t)N * h * w;
    const size_t plane = (size_t)h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = i % w;
        const size_t r = i / w;
        const int y = r % h;
        const size_t n = r / h;
        const float a = yl[i];
        const float* hb = yh + n * 3 * plane + (size_t)y * w + x;
        const float b = hb[0], c = hb[plane], d = hb[2 * plane];
        float v[4] = {(a + b + c + d) * 0.5f, (a + b - c - d) * 0.5f, (a - b + c - d) * 0.5f, (a - b - c + d) * 0.5f};
        const size_t dst = n * 4 * plane + (size_t)(2 * y) * (2 * w) + 2 * x;
        const size_t o[4] = {dst, dst + 1, dst + 2 * w, dst + 2 * w + 1};
        for (int k = 0; k < 4; ++k) {
            out[o[k]] = v[k];
            if (disp) {
                float t = v[k] * disp_scale;
                if (clamp01) t = fminf(fmaxf(t, 0.f), 1.f);
                disp[o[k]] = t;
            }
        }
    }
}

// g = d_out + d_disp * scale * [0 <= out*scale <= 1]; then the analysis butterfly.
// Also used (d_disp == nullptr) as the forward DWT.
__global__ void haar_analysis_kernel(const float* __restrict__ g_out, const float* __restrict__ g_disp,
                                     const float* __restrict__ out, float* __restrict__ yl, float* __restrict__ yh,
                                     int N, int h, int w, float disp_scale, int clamp01) {
    const size_t total = (size_t)N * h * w;
    const size_t plane = (size_t)h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = i % w;
        const size_t r = i / w;
        const int y = r % h;
        const size_t n = r / h;
        const size_t src = n * 4 * plane + (size_t)(2 * y) * (2 * w) + 2 * x;
        float2 t = make_float2(0.f, 0.f), u = make_float2(0.f, 0.f);
        if (g_out) {
            t = *reinterpret_cast<const float2*>(g_out + src);
            u = *reinterpret_cast<const float2*>(g_out + src + 2 * w);
        }
        if (g_disp) {
            float2 dt = *reinterpret_cast<const float2*>(g_disp + src);
            float2 du = *reinterpret_cast<const float2*>(g_disp + src + 2 * w);
            if (clamp01) {
                // torch.clamp backward passes the gradient where min <= x <= max (inclusive)
                const float2 ot = *reinterpret_cast<const float2*>(out + src);
                const float2 ou = *reinterpret_cast<const float2*>(out + src + 2 * w);
                float s;
                s = ot.x * disp_scale; if (!(s >= 0.f && s <= 1.f)) dt.x = 0.f;
                s = ot.y * disp_scale; if (!(s >= 0.f && s <= 1.f)) dt.y = 0.f;
                s = ou.x * disp_scale; if (!(s >= 0.f && s <= 1.f)) du.x = 0.f;
                s = ou.y * disp_scale; if (!(s >= 0.f && s <= 1.f)) du.y = 0.f;
            }
            t.x += dt.x * disp_scale; t.y += dt.y * disp_scale;
            u.x += du.x * disp_scale; u.y += du.y * disp_scale;
        }
        yl[i] = (t.x + t.y + u.x + u.y) * 0.5f;
        float* hb = yh + n * 3 * plane + (size_t)y * w + x;
        hb[0] = (t.x + t.y - u.x - u.y) * 0.5f;
        hb[plane] = (t.x - t.y + u.x - u.y) * 0.5f;
        hb[2 * plane] = (t.x - t.y - u.x + u.y) * 0.5f;
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz,
                               size_t n, int act, float slope) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = dy[i], v = y[i];
        const float d = act_deriv(v, act, slope);
        dz[i] = g * d;
    }
}

static inline int grid_for(size_t work, int block) {
    size_t b = (work + block - 1) / block;
    return (int)std::min<size_t>(std::max<size_t>(b, 1), (size_t)kNumCU * 8);
}

}  // namespace wmd

using namespace wmd;

extern "C" int wmd_idwt_haar_fwd(const float* yl, const float* yh, float* out, float* disp, int N, int h, int w,
                                 float disp_scale, int clamp01, void* stream) {
    if (!yl || !yh || !out) return fail(WMD_ERR_BAD_ARG, "wmd_idwt_haar_fwd: null pointer");
    if (N < 0 || h <= 0 || w <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_idwt_haar_fwd: N=%d h=%d w=%d", N, h, w);
    if (N == 0) return WMD_OK;
    // 8 B moved and 4 FLOP per output pixel (reference op model depth_decoder.py:373), 12 B with disp
    ProfScope prof("idwt_haar_fwd_kernel", 16.0 * N * h * w, (disp ? 48.0 : 32.0) * N * h * w, (hipStream_t)stream);
    if ((w & 1) == 0) {
        const size_t work = (size_t)N * h * (w / 2);
        hipLaunchKernelGGL(idwt_haar_fwd_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, yl, yh,
                           out, disp, N, h, w, disp_scale, clamp01);
    } else {
        const size_t work = (size_t)N * h * w;
        hipLaunchKernelGGL(idwt_haar_fwd_scalar_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream,
                           yl, yh, out, disp, N, h, w, disp_scale, clamp01);
    }
    return check_launch("idwt_haar_fwd_kernel");
}

extern "C" int wmd_idwt_haar_bwd(const float* d_out, const float* d_disp, const float* out, float* d_yl, float* d_yh,
                                 int N, int h, int w, float disp_scale, int clamp01, void* stream) {
    if (!d_yl || !d_yh) return fail(WMD_ERR_BAD_ARG, "wmd_idwt_haar_bwd: null output pointer");
    if (d_disp && clamp01 && !out) return fail(WMD_ERR_BAD_ARG, "wmd_idwt_haar_bwd: clamp mask needs `out`");
    if (N < 0 || h <= 0 || w <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_idwt_haar_bwd: N=%d h=%d w=%d", N, h, w);
    if (N == 0) return WMD_OK;
    const size_t work = (size_t)N * h * w;
    ProfScope prof("haar_analysis_kernel", 16.0 * work, (d_out && d_disp ? 64.0 : 32.0) * work, (hipStream_t)stream);
    hipLaunchKernelGGL(haar_analysis_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, d_out,
                       d_disp, out, d_yl, d_yh, N, h, w, disp_scale, clamp01);
    return check_launch("haar_analysis_kernel");
}

extern "C" int wmd_dwt_haar_fwd(const float* x, float* yl, float* yh, int N, int h, int w, void* stream) {
    if (!x || !yl || !yh) return fail(WMD_ERR_BAD_ARG, "wmd_dwt_haar_fwd: null pointer");
    if (N < 0 || h <= 0 || w <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_dwt_haar_fwd: N=%d h=%d w=%d", N, h, w);
    if (N == 0) return WMD_OK;
    const size_t work = (size_t)N * h * w;
    ProfScope prof("haar_analysis_kernel", 16.0 * work, 32.0 * work, (hipStream_t)stream);
    hipLaunchKernelGGL(haar_analysis_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (const float*)nullptr, (const float*)nullptr, yl, yh, N, h, w, 1.f, 0);
    return check_launch("haar_analysis_kernel");
}

// DWT(J, "haar", mode="reflect") on ANY size: pytorch_wavelets pads an odd axis by one reflected sample on the right / bottom
// (afb1d: p = 2*(ceil(N/2) - 1) - N + 2 = 1 -> F.pad(x, (0, 1), "reflect"), i.e. x[N] = x[N-2]) before the stride-2 filter
// pair.  Folded into the load indices here.
__global__ void haar_analysis_reflect_kernel(const float* __restrict__ x, float* __restrict__ yl, float* __restrict__ yh,
                                             int N, int H, int W, int h, int w) {
    const size_t total = (size_t)N * h * w;
    const size_t plane = (size_t)h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = i % w;
        const size_t r = i / w;
        const int y = r % h;
        const size_t n = r / h;
        const int y0 = 2 * y, x0 = 2 * xx;
        const int y1 = y0 + 1 < H ? y0 + 1 : H - 2, x1 = x0 + 1 < W ? x0 + 1 : W - 2;
        const float* p = x + n * (size_t)H * W;
        const float a = p[(size_t)y0 * W + x0], b = p[(size_t)y0 * W + x1], c = p[(size_t)y1 * W + x0], d = p[(size_t)y1 * W + x1];
        yl[i] = (a + b + c + d) * 0.5f;
        float* hb = yh + n * 3 * plane + (size_t)y * w + xx;
        hb[0] = (a + b - c - d) * 0.5f;
        hb[plane] = (a - b + c - d) * 0.5f;
        hb[2 * plane] = (a - b - c + d) * 0.5f;
    }
}

extern "C" int wmd_dwt_haar_reflect_fwd(const float* x, float* yl, float* yh, int N, int H, int W, void* stream) {
    if (!x || !yl || !yh) return fail(WMD_ERR_BAD_ARG, "wmd_dwt_haar_reflect_fwd: null pointer");
    if (N < 0 || H <= 0 || W <= 0) return fail(WMD_ERR_BAD_SHAPE, "wmd_dwt_haar_reflect_fwd: N=%d H=%d W=%d", N, H, W);
    if (((H & 1) && H < 2) || ((W & 1) && W < 2))
        return fail(WMD_ERR_BAD_SHAPE, "wmd_dwt_haar_reflect_fwd: reflection padding of an odd axis needs >= 2 samples (%dx%d)", H, W);
    if (N == 0) return WMD_OK;
    const int h = (H + 1) / 2, w = (W + 1) / 2;
    const size_t work = (size_t)N * h * w;
    ProfScope prof("haar_analysis_reflect_kernel", 16.0 * work, 32.0 * work, (hipStream_t)stream);
    hipLaunchKernelGGL(haar_analysis_reflect_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, x, yl, yh, N, H, W, h, w);
    return check_launch("haar_analysis_reflect_kernel");
}

extern "C" int wmd_act_bwd(const float* dy, const float* y, float* dz, size_t n, int act, float slope, void* stream) {
    if (!dy || !y || !dz) return fail(WMD_ERR_BAD_ARG, "wmd_act_bwd: null pointer");
    if (act < 0 || act > 3) return fail(WMD_ERR_BAD_ARG, "wmd_act_bwd: act=%d", act);
    if (n == 0) return WMD_OK;
    ProfScope prof("act_bwd_kernel", (double)n, 12.0 * n, (hipStream_t)stream);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dz, n, act,
                       slope);
    return check_launch("act_bwd_kernel");
}
