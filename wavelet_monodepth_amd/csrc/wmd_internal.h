// Internal helpers shared by the HIP translation units of libwmd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/wmd.h"

namespace wmd {

void set_error(const char* fmt, ...);

inline int fail(int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return status;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(WMD_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return WMD_OK;
}

constexpr int kNumCU = 256;  // MI355X

// Opt-in per-launch timing (wmd_profile_begin/end). Zero cost when off: one predictable branch.
extern bool g_prof_on;
int prof_open(const char* name, double flops, double bytes, hipStream_t s);
void prof_close(int idx, hipStream_t s);
void prof_set_mfma(int idx, double mfma_flops);   // FLOPs the matrix pipe executes for this launch (default: the algorithmic count)
struct ProfScope {
    int idx;
    hipStream_t s;
    ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : idx(-1), s(stream) {
        if (g_prof_on) idx = prof_open(name, flops, bytes, stream);
    }
    void mfma(double f) {
        if (idx >= 0) prof_set_mfma(idx, f);
    }
    ~ProfScope() {
        if (idx >= 0) prof_close(idx, s);
    }
};

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case WMD_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case WMD_ACT_LEAKY: return v > 0.f ? v : v * slope;
        case WMD_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// f'(x) written in terms of the activation OUTPUT y = f(x)
__device__ __forceinline__ float act_deriv(float y, int act, float slope) {
    switch (act) {
        case WMD_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;     // y = e^x - 1  =>  dy/dx = y + 1
        case WMD_ACT_LEAKY: return y > 0.f ? 1.f : slope;
        case WMD_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// map a padded coordinate g in [-1, n] to a source coordinate; returns false when the tap reads zero
__device__ __forceinline__ bool pad_coord(int& g, int n, int pad_mode) {
    if (pad_mode == WMD_PAD_REFLECT) {
        if (g < 0) g = -g;
        if (g >= n) g = 2 * n - 2 - g;
    } else if (pad_mode == WMD_PAD_REPLICATE) {
        g = g < 0 ? 0 : (g >= n ? n - 1 : g);
    } else {
        if (g < 0 || g >= n) return false;
    }
    return true;
}

// wmd_head_chain.hip: the chained-GEMM form of wmd_head_fused_fwd (chain = 0); false = not taken (unsupported shape / switched off)
// -> 0 not taken, 1 taken, 2 taken together with the low-pass chain (wmd_head_fused_args.ll_wp1)
int head_chain_launch(const wmd_head_fused_args* g, int t_planes, hipStream_t s);
// round 6: the chained first stages of up to three levels in ONE launch (0 = not taken; wmd_head_fused_multi_fwd)
int head_chain_multi_launch(const wmd_head_fused_args* levels, int n, hipStream_t s);

// wmd_head_stream.hip: the streaming form of wmd_head_level_fwd (C = 32, plain inference outputs); 0 = not taken
int head_stream_launch(const wmd_head_level_args* g, const wmd_head_shiftsum_args* coarse, int n_coarse, hipStream_t s);
int head_stream_pyramid_pays(int B, int H, int W);

}  // namespace wmd
