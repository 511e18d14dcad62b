// Backward of the 1x1 stage of a level's wavelet heads (training path) for gfx950.
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
#include "wmd_head_bwd1.h"

namespace wmd {

__global__ __launch_bounds__(256) void head1x1_bwd_data_kernel(const Head1x1K a) { head1x1_bwd_data_body(a, blockIdx.x, blockIdx.y, gridDim.x); }

__global__ __launch_bounds__(H1_WW * 64) void head1x1_bwd_weight_kernel(const Head1x1K a) {
    __shared__ __attribute__((aligned(16))) float smem[H1_SMEM_FLOATS];
    head1x1_bwd_weight_body(a, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, smem);
}

__global__ __launch_bounds__(256) void head1x1_bwd_reduce_kernel(const Head1x1K a) { head1x1_bwd_reduce_body(a, blockIdx.x, blockIdx.y, gridDim.x); }

int head1x1_blocks(const wmd_head1x1_bwd_args* g) {
    const long tiles = (long)g->B * (((long)g->H * g->W + 63) / 64);
    return (int)std::max<long>(1, std::min<long>((tiles + H1_WW - 1) / H1_WW, H1_MAX_BLK));
}

}  // namespace wmd

using namespace wmd;

namespace wmd {
int head1x1_validate(const wmd_head1x1_bwd_args* g) {
    if (!g) return fail(WMD_ERR_BAD_ARG, "wmd_head1x1_bwd: null args");
    if (!g->dz || !g->x || !g->w1 || !g->dw1 || !g->db1) return fail(WMD_ERR_BAD_ARG, "wmd_head1x1_bwd: null tensor pointer");
    if (g->B <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->Ct <= 0)
        return fail(WMD_ERR_BAD_SHAPE, "wmd_head1x1_bwd: B=%d H=%d W=%d C=%d Ct=%d", g->B, g->H, g->W, g->C, g->Ct);
    if (g->Ct % 8) return fail(WMD_ERR_UNSUPPORTED, "wmd_head1x1_bwd: Ct=%d is not a multiple of 8", g->Ct);
    if (g->x_act != WMD_ACT_NONE && g->x_act != WMD_ACT_LEAKY && g->x_act != WMD_ACT_ELU)
        return fail(WMD_ERR_UNSUPPORTED, "wmd_head1x1_bwd: x_act=%d (none, LeakyReLU or ELU)", g->x_act);
    return WMD_OK;
}

void head1x1_fill(const wmd_head1x1_bwd_args* g, Head1x1K* a) {
    memset(a, 0, sizeof(*a));
    a->dz = g->dz, a->x = g->x, a->w1 = g->w1, a->dx = g->dx, a->dw1 = g->dw1, a->db1 = g->db1;
    a->B = g->B, a->HW = g->H * g->W, a->C = g->C, a->Ct = g->Ct;
    a->dslope = g->x_act == WMD_ACT_LEAKY ? g->x_slope : 1.f;
    a->delu = g->x_act == WMD_ACT_ELU ? 1.f : 0.f;
    a->nblk = head1x1_blocks(g);
    a->n_cgrp = (g->Ct + 63) / 64;
    a->n_igrp = (g->C + 63) / 64;
    a->partial = g->workspace;
}
}  // namespace wmd

extern "C" size_t wmd_head1x1_bwd_workspace_floats(const wmd_head1x1_bwd_args* g) {
    if (head1x1_validate(g)) return 0;
    return (size_t)head1x1_blocks(g) * ((g->Ct + 63) / 64) * ((g->C + 63) / 64) * H1_PART;
}

extern "C" int wmd_head1x1_bwd(const wmd_head1x1_bwd_args* g, void* stream) {
    if (int st = head1x1_validate(g)) return st;
    Head1x1K a;
    head1x1_fill(g, &a);
    const size_t need = (size_t)a.nblk * a.n_cgrp * a.n_igrp * H1_PART;
    if (!g->workspace || g->workspace_floats < need)
        return fail(WMD_ERR_WORKSPACE, "wmd_head1x1_bwd: workspace %zu < %zu floats", g->workspace_floats, need);
    hipStream_t s = (hipStream_t)stream;
    const double pix = (double)g->B * g->H * g->W;
    const long tiles = (long)g->B * (((long)g->H * g->W + 63) / 64);
    if (g->dx) {
        ProfScope prof("head1x1_bwd_data_kernel", 2.0 * g->C * g->Ct * pix, 4.0 * pix * (g->Ct + 2.0 * g->C), s);
        const dim3 grid((unsigned)std::max<long>(1, std::min<long>((tiles + 3) / 4, 4096)), a.n_igrp);
        hipLaunchKernelGGL(head1x1_bwd_data_kernel, grid, dim3(256), 0, s, a);
        if (int st = check_launch("head1x1_bwd_data_kernel")) return st;
    }
    {
        ProfScope prof("head1x1_bwd_weight_kernel", 2.0 * g->C * g->Ct * pix, 4.0 * pix * (g->Ct + g->C), s);
        hipLaunchKernelGGL(head1x1_bwd_weight_kernel, dim3(a.nblk, a.n_cgrp * a.n_igrp), dim3(H1_WW * 64), 0, s, a);
    }
    if (int st = check_launch("head1x1_bwd_weight_kernel")) return st;
    {
        ProfScope prof("head1x1_bwd_reduce_kernel", (double)need, 4.0 * need, s);
        hipLaunchKernelGGL(head1x1_bwd_reduce_kernel, dim3(a.n_cgrp * a.n_igrp, (64 * 64 + 64 + 15) / 16), dim3(256), 0, s, a);
    }
    return check_launch("head1x1_bwd_reduce_kernel");
}
