// Wavelet-coefficient heads: 3x3 convolutions with 1..4 output channels (HBM/LDS-bound, VALU).
//
// Replaces the trailing Conv3x3(C,3|1) + Sigmoid of the `waveconv` heads and the
// 2^(s-1)*(sigmoid(+) - sigmoid(-)) combine of get_coefficients
// (KITTI/networks/decoders/depth_decoder.py:104-136), and the NYUv2 wave1_ll / wave{1,2,3}
// convolutions (NYUv2/networks/decoders/densedepth_decoder.py:106-115,122-141).
//
// With <= 4 output channels an MFMA tile would be > 75 % padding, so this is a direct convolution:
// one output pixel per thread, 8x32 pixel tiles, CK-channel halo patches staged through LDS and the
// filter taps read through the scalar cache (wave-uniform addresses).  Output goes straight into the
// [B,3,H,W] coefficient plane that the IDWT reads, so the sigmoid/scale/subtract never touch HBM.
#include <algorithm>
#include "wmd_internal.h"

namespace wmd {

constexpr int HT_H = 8, HT_W = 32, H_CK = 8;
constexpr int H_PH = HT_H + 2, H_PW = HT_W + 2;
constexpr int H_PS = H_PH * H_PW + 1;  // odd stride: channel planes start on different banks

template <int COUT>
__device__ __forceinline__ void head_accumulate(const float* __restrict__ x, const float* __restrict__ wgt, int C,
                                                int H, int W, int b, int y0, int x0, int pad_mode, float* lds,
                                                float (&acc)[COUT]) {
    const int tid = threadIdx.x;
    const int ly = tid / HT_W, lx = tid % HT_W;
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * C * plane;

    // staging geometry (positions tid and tid+256 of the 10x34 patch)
    int off[2];
    bool live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = tid + i * 256;
        int gy = y0 + p / H_PW - 1, gx = x0 + p % H_PW - 1;
        bool ok = p < H_PH * H_PW;
        ok = pad_coord(gy, H, pad_mode) && ok;
        ok = pad_coord(gx, W, pad_mode) && ok;
        ok = ok && gy >= 0 && gx >= 0 && gy < H && gx < W;
        gy = min(max(gy, 0), H - 1);
        gx = min(max(gx, 0), W - 1);
        live[i] = ok;
        off[i] = gy * W + gx;
    }

    for (int c0 = 0; c0 < C; c0 += H_CK) {
        __syncthreads();  // previous chunk (or previous side) fully consumed
#pragma unroll
        for (int j = 0; j < H_CK; ++j) {
            const int ci = c0 + j;
            const float* src = xb + (size_t)ci * plane;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = tid + i * 256;
                if (p < H_PH * H_PW) lds[j * H_PS + p] = (ci < C && live[i]) ? src[off[i]] : 0.f;
            }
        }
        __syncthreads();
        const int nj = min(H_CK, C - c0);
        for (int j = 0; j < nj; ++j) {
            const float* pt = lds + j * H_PS + ly * H_PW + lx;
            float v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = pt[(t / 3) * H_PW + (t % 3)];
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float* wk = wgt + ((size_t)co * C + (c0 + j)) * 9;  // wave-uniform -> scalar loads
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[co] = fmaf(wk[t], v[t], acc[co]);
            }
        }
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void head3x3_kernel(const wmd_head_args a, int tiles_x, int tiles_y) {
    __shared__ float lds[H_CK * H_PS];
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * HT_H, x0 = tx * HT_W;
    const int oy = y0 + threadIdx.x / HT_W, ox = x0 + threadIdx.x % HT_W;
    const bool ok = oy < a.H && ox < a.W;

    float accp[COUT], accn[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        accp[co] = a.bias_p ? a.bias_p[co] : 0.f;
        accn[co] = 0.f;
    }
    head_accumulate<COUT>(a.xp, a.wgt_p, a.C, a.H, a.W, b, y0, x0, a.pad_mode, lds, accp);
    if (a.mode == 2) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) accn[co] = a.bias_n ? a.bias_n[co] : 0.f;
        head_accumulate<COUT>(a.xn, a.wgt_n, a.C, a.H, a.W, b, y0, x0, a.pad_mode, lds, accn);
    }
    if (!ok) return;
    const size_t plane = (size_t)a.H * a.W;
    const size_t o = (size_t)b * COUT * plane + (size_t)oy * a.W + ox;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float r;
        if (a.mode == 0) {
            r = a.scale * accp[co];
        } else {
            const float sp = 1.f / (1.f + expf(-accp[co]));
            if (a.sig_p) a.sig_p[o + co * plane] = sp;
            if (a.mode == 1) {
                r = a.scale * sp;
            } else {
                const float sn = 1.f / (1.f + expf(-accn[co]));
                if (a.sig_n) a.sig_n[o + co * plane] = sn;
                // reference order (depth_decoder.py:134-135): 2^(s-1)*sig(+) - 2^(s-1)*sig(-)
                r = a.scale * sp - a.scale * sn;
            }
        }
        a.y[o + co * plane] = r;
    }
}

}  // namespace wmd

using namespace wmd;

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
    const int tiles_x = (g->W + HT_W - 1) / HT_W, tiles_y = (g->H + HT_H - 1) / HT_H;
    dim3 grid((unsigned)((size_t)g->B * tiles_x * tiles_y));
    hipStream_t s = (hipStream_t)stream;
    const double pix = (double)g->B * g->H * g->W, sides = g->mode == 2 ? 2.0 : 1.0;
    ProfScope prof("head3x3_kernel", sides * 18.0 * g->C * g->Cout * pix, 4.0 * pix * (sides * g->C + g->Cout), s);
    switch (g->Cout) {
        case 1: hipLaunchKernelGGL(head3x3_kernel<1>, grid, dim3(256), 0, s, *g, tiles_x, tiles_y); break;
        case 2: hipLaunchKernelGGL(head3x3_kernel<2>, grid, dim3(256), 0, s, *g, tiles_x, tiles_y); break;
        case 3: hipLaunchKernelGGL(head3x3_kernel<3>, grid, dim3(256), 0, s, *g, tiles_x, tiles_y); break;
        default: hipLaunchKernelGGL(head3x3_kernel<4>, grid, dim3(256), 0, s, *g, tiles_x, tiles_y); break;
    }
    return check_launch("head3x3_kernel");
}
