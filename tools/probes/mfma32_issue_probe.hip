// What does a wave pay for the instructions BETWEEN its v_mfma_f32_32x32x2_f32 (round 5)?  One loop per variant:
//   NV independent VALU adds producing the next A operand + NL ds_read_b32 (weight-fragment-like, prefetched 3 deep) per MFMA,
//   8 accumulators in rotation, sched_barrier-pinned like conv_wino32_kernel; 1, 2 or 3 waves per SIMD (blocks of 256 threads,
//   one wave per SIMD each; co-residency forced by the grid: blocks = CUs x waves-per-SIMD, LDS sized so that they fit).
// Prints cycles per MFMA and the fraction of the 64-cycle issue rate.   hipcc --offload-arch=gfx950 -O3 -o mfma32_issue_probe ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NL>
__global__ __launch_bounds__(256, 3) void probe(const float* __restrict__ x, float* out, int iters) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = x[i];
    __syncthreads();
    f32x16 acc[8];
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = x[lane + i * 64];
    float wf[4] = {lds[lane], lds[lane + 64], lds[lane + 128], lds[lane + 192]};
    const float* wsrc = lds + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (NL >= 1) wf[(s + 3) & 3] = wsrc[((it * 8 + s + 3) & 31) * 64];
            if (NL >= 2) v[15] += wsrc[((it * 8 + s) & 31) * 64 + 2048];
            float a = v[s & 7];
#pragma unroll
            for (int k = 0; k < NV; ++k) {   // a chain of length 2 per pair: independent of the MFMA results
                v[(s + k) & 7] = v[(s + k) & 7] + v[8 + (k & 7)];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wf[s & 3], acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0.f;
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) sum += acc[o][r];
    for (int i = 0; i < 16; ++i) sum += v[i];
    out[blockIdx.x * 256 + tid] = sum;
}

// the same loop with the NV adds done as NV / 2 v_pk_add_f32 (two fp32 adds per lane per instruction)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NV, int NL>
__global__ __launch_bounds__(256, 3) void probe_pk(const float* __restrict__ x, float* out, int iters) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = x[i];
    __syncthreads();
    f32x16 acc[8];
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{x[lane + i * 128], x[lane + i * 128 + 64]};
    float wf[4] = {lds[lane], lds[lane + 64], lds[lane + 128], lds[lane + 192]};
    const float* wsrc = lds + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (NL >= 1) wf[(s + 3) & 3] = wsrc[((it * 8 + s + 3) & 31) * 64];
            if (NL >= 2) v[7][1] += wsrc[((it * 8 + s) & 31) * 64 + 2048];
            float a = v[(s >> 1) & 3][s & 1];
#pragma unroll
            for (int k = 0; k < NV / 2; ++k)   // inline asm: the compiler's post-RA peephole unpacks a plain f32x2 add behind an MFMA
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v[(s + k) & 3]) : "v"(v[(s + k) & 3]), "v"(v[4 + (k & 3)]));
            __builtin_amdgcn_sched_barrier(0);
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wf[s & 3], acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0.f;
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) sum += acc[o][r];
    for (int i = 0; i < 8; ++i) sum += v[i][0] + v[i][1];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int NV, int NL, bool PK = false>
void run(int wps, const float* x, float* out) {
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto kern = PK ? probe_pk<NV, NL> : probe<NV, NL>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wps waves x iters x 8 MFMAs of 4096 FLOP
    const double mfmas_per_simd = (double)wps * iters * 8;
    const double tflops = 1024.0 * mfmas_per_simd * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%s VALU/MFMA %2d  ds_read/MFMA %d  waves/SIMD %d : %7.1f us  %6.1f TFLOP/s = %.3f of 157.3 -> %.0f core cycles per MFMA per SIMD at 2.4 GHz\n",
           PK ? "packed" : "plain ", NV, NL, wps, ms * 1e3, tflops, tflops / 157.3, ms * 1e-3 * 2.4e9 / mfmas_per_simd);
}

int main() {
    float *x, *out;
    hipMalloc(&x, 1 << 20); hipMalloc(&out, 64 << 20);
    hipMemset(x, 0, 1 << 20);
    for (int wps = 1; wps <= 3; ++wps) {
        run<0, 0>(wps, x, out); run<0, 1>(wps, x, out); run<2, 1>(wps, x, out); run<4, 1>(wps, x, out); run<4, 2>(wps, x, out);
        run<8, 1>(wps, x, out); run<8, 2>(wps, x, out); run<16, 2>(wps, x, out);
        run<2, 1, true>(wps, x, out); run<4, 1, true>(wps, x, out); run<4, 2, true>(wps, x, out); run<8, 2, true>(wps, x, out);
    }
    return 0;
}
