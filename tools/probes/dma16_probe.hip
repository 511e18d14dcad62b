// Does buffer_load ... lds with 16-byte pieces accept global addresses that are only 4-byte aligned (gfx950)?  And what do the
// per-dword range checks of a raw buffer do to a piece that straddles num_records?  (development probe for the trunk's staging)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, lds_ptr_t dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void lds_dma4(__amdgpu_buffer_rsrc_t r, lds_ptr_t dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 4, voff, soff, 0, 0);
}
// mode 0: x4 pieces, lane l loads 16 B from byte offset shift*4 + l*stride*4 ; mode 1: the same data with dword pieces
__global__ void probe(const float* src, float* out, int n, int shift, int stride, int mode, int reps) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 8];
    const int lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * 4, 0x00020000);
    for (int it = 0; it < reps; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p) lds_dma16(r, (lds_ptr_t)(lds + p * 256), (unsigned)((shift + (lane + 64 * p) * stride) * 4), 0);
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // dword pieces write 64 consecutive dwords: element e = k*64 + lane of piece p  <->  group g = e / 4, word e % 4
                    const int e = k * 64 + lane, g = e >> 2, w = e & 3;
                    lds_dma4(r, (lds_ptr_t)(lds + p * 256 + k * 64), (unsigned)((shift + (g + 64 * p) * stride + w) * 4), 0);
                }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    for (int i = lane; i < 64 * 4 * 8; i += 64) out[i] = lds[i];
}
int main() {
    const int n = 64 * 8 * 40 + 64;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4);
    hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> res(2048);
    for (int shift = 0; shift < 4; ++shift)
        for (int stride : {4, 9, 36}) {
            int bad[2] = {0, 0};
            for (int mode = 0; mode < 2; ++mode) {
                hipMemset(o, 0, 2048 * 4);
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, n, shift, stride, mode, 1);
                hipMemcpy(res.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
                for (int g = 0; g < 512; ++g)
                    for (int w = 0; w < 4; ++w) {
                        const int idx = shift + g * stride + w;
                        const float want = idx < n ? (float)idx : 0.f;
                        if (res[g * 4 + w] != want) ++bad[mode];
                    }
            }
            printf("shift %d (byte offset %% 16 = %2d) stride %2d dwords: x4 pieces %s (%d wrong), dword pieces %s (%d wrong)\n", shift, (shift * 4) % 16,
                   stride, bad[0] ? "WRONG" : "ok", bad[0], bad[1] ? "WRONG" : "ok", bad[1]);
        }
    // timing: 64 blocks x 64 lanes x reps, x4 vs dword
    for (int mode = 0; mode < 2; ++mode)
        for (int shift : {0, 3}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, d, o, n, shift, 9, mode, 200);
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, d, o, n, shift, 9, mode, 2000);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d (%s) shift %d: %.1f us for 2000 x 8 KB per wave, 1024 waves -> %.1f ns per 8 KB wave-iteration\n", mode, mode ? "dword" : "x4", shift,
                   ms * 1e3, ms * 1e6 / 2000);
        }
    return 0;
}
