// One coefficient pixel of a level's completion (nine-tap gathers of the tap-partial planes, bias, sigmoid, combine, Haar butterfly):
// shared by head_shiftsum_chain_kernel (wmd_head.hip) and by the pyramid that head_stream_kernel's epilogue waves run (round 6,
// wmd_head_stream.hip), so that every form of the completion produces the same bits.
#pragma once
#include "wmd_internal.h"

namespace wmd {

// sigmoid of the completion kernels (round 6): v_exp_f32 + v_rcp_f32 (1 ulp each, ~2e-7 on the value) instead of expf + an IEEE
// division (~60 instructions for 7 sigmoids per pixel in kernels that are made of latency and instruction issue); the per-level and
// the chained completion share it, so their outputs stay bit-identical to each other
__device__ __forceinline__ float fast_sigmoid(float h) { return __builtin_amdgcn_rcpf(1.f + __expf(-h)); }

// One coefficient pixel of one level in two parts, so that a caller may issue the gathers of several levels at once and keep only the
// low-pass hand-over sequential (head_stream_kernel's pyramid):
//   shiftsum_gather: nine-tap gathers, bias, sigmoid, combine -> yh[3] (stored) and, when the level owns its low-pass value (the
//                    low-pass head's planes 54..62, or a.yl), that value (stored to yl_out); returns whether l was set
//   shiftsum_finish: the Haar butterfly of (l, yh) -> v[4], stored to out / disp when the level has them
__device__ __forceinline__ bool shiftsum_gather(const wmd_head_shiftsum_args& a, size_t b, int y, int x, float (&yh)[3], float& l) {
    const int H = a.H, W = a.W;
    const size_t plane = (size_t)H * W, i = b * plane + (size_t)y * W + x;
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
        yh[co] = a.scale * a1 - a.scale * a2;
        a.yh[(b * 3 + co) * plane + (size_t)y * W + x] = yh[co];
    }
    if (a.yl_out) {
        float sl = a.bias_ll ? a.bias_ll[0] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) sl += okf[t] * tb[(size_t)(54 + t) * plane + off[t]];
        l = a.scale_ll * fast_sigmoid(sl);
        a.yl_out[i] = l;
        return true;
    }
    if (a.yl) {
        l = a.yl[i];
        return true;
    }
    return false;
}

__device__ __forceinline__ void shiftsum_finish(const wmd_head_shiftsum_args& a, size_t b, int y, int x, const float (&yh)[3], float l, float (&v)[4]) {
    const int W = a.W;
    const size_t plane = (size_t)a.H * W;
    v[0] = (l + yh[0] + yh[1] + yh[2]) * 0.5f;
    v[1] = (l + yh[0] - yh[1] - yh[2]) * 0.5f;
    v[2] = (l - yh[0] + yh[1] - yh[2]) * 0.5f;
    v[3] = (l - yh[0] - yh[1] + yh[2]) * 0.5f;
    if (a.out) {
        const size_t dst = b * 4 * plane + (size_t)(2 * y) * (2 * W) + 2 * x;
        *reinterpret_cast<float2*>(a.out + dst) = make_float2(v[0], v[1]);
        *reinterpret_cast<float2*>(a.out + dst + 2 * W) = make_float2(v[2], v[3]);
        if (a.disp) {
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = v[k] * a.disp_scale;
                if (a.clamp01) d[k] = fminf(fmaxf(d[k], 0.f), 1.f);
            }
            *reinterpret_cast<float2*>(a.disp + dst) = make_float2(d[0], d[1]);
            *reinterpret_cast<float2*>(a.disp + dst + 2 * W) = make_float2(d[2], d[3]);
        }
    }
}

// both parts in sequence (head_shiftsum_chain_kernel): l = the level's own low-pass value, else l_in (have_l_in), else 0
__device__ __forceinline__ void shiftsum_pixel(const wmd_head_shiftsum_args& a, size_t b, int y, int x, bool have_l_in, float l_in, float (&v)[4]) {
    float yh[3], l = 0.f;
    const bool own = shiftsum_gather(a, b, y, x, yh, l);
    if (!own) l = have_l_in ? l_in : 0.f;
    shiftsum_finish(a, b, y, x, yh, l, v);
}

}  // namespace wmd
