// What would fewer LDS weight-fragment instructions buy conv_wino32_kernel's main loop (round 5)?  A replica of one K-step of the
// half-position wave after the packed transform -- per 8 MFMAs 32x32x2: three 16-byte patch rows (ds_read2_b64), ten v_pk_add_f32,
// and the weight fragments read as   WMODE 0: eight ds_read_b32 (the kernel today)   1: four ds_read2_b32   2: two ds_read_b128
//                                    3: none (bound)        PK 0: no transform arithmetic (bound)
// 1, 2 or 3 waves per SIMD, zero data (so the part holds its top clock: fractions are of the matrix pipe, not of a power budget).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int WMODE, int PK>
__global__ __launch_bounds__(256, 3) void probe(const float* __restrict__ x, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = x[i];
    __syncthreads();
    f32x16 acc[8];
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;
    f32x2 vp[2][4];
    for (int i = 0; i < 4; ++i) vp[0][i] = vp[1][i] = f32x2{x[lane + i * 128], x[lane + i * 128 + 64]};
    float wf[16];
    for (int i = 0; i < 16; ++i) wf[i] = lds[lane + 64 * i];
    const float* psrc0 = lds + 4096 + (lane & 31) * 2;
    const float* wsrc0 = lds + lane * (WMODE == 2 ? 4 : 1);
    for (int it4 = 0; it4 < iters; it4 += 4) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int it = kk;
        const float* psrc = psrc0 + (it4 & 4) * 160;   // varies: nothing is hoisted out of the loop
        const float* wsrc = wsrc0 + (it4 & 4) * 8;
        // patch rows of the next K-step
        f32x2 dp[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            dp[2 * r] = *reinterpret_cast<const f32x2*>(psrc + kk * 160 + r * 36);
            dp[2 * r + 1] = *reinterpret_cast<const f32x2*>(psrc + kk * 160 + r * 36 + 2);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (WMODE == 0) wf[(s + 3) & 7] = wsrc[(kk * 8 + ((s + 3) & 7)) * 64];
            if (WMODE == 1 && (s & 1) == 0) {
                wf[(s + 4) & 7] = wsrc[(kk * 4 + (s >> 1)) * 64];
                wf[8 + ((s + 4) & 7)] = wsrc[(kk * 4 + (s >> 1)) * 64 + 32];
            }
            if (WMODE == 2 && (s & 3) == 0) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wsrc + (kk * 2 + (s >> 2)) * 256);
                wf[((s + 4) & 7) + 0] = w4[0], wf[((s + 4) & 7) + 1] = w4[1], wf[((s + 4) & 7) + 2] = w4[2], wf[((s + 4) & 7) + 3] = w4[3];
            }
            if (PK) {
                if (s == 2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 a = pk_add(dp[h], dp[2 + h]), b = pk_add(dp[2 + h], dp[4 + h]), c = pk_add(dp[h], dp[4 + h]);
                        dp[h] = a, dp[2 + h] = b, dp[4 + h] = c;
                    }
                }
                if (s == 3) {
                    f32x2* vv = vp[(it + 1) & 1];
                    vv[0] = pk_add(dp[0], dp[1]), vv[1] = pk_add(dp[2], dp[3]), vv[2] = pk_add(dp[4], dp[5]), vv[3] = pk_add(dp[0], dp[5]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[it & 1][s >> 1][s & 1], WMODE == 1 ? wf[(s & 1) * 8 + (s & 6)] : wf[s], acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    float sum = 0.f;
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) sum += acc[o][r];
    for (int i = 0; i < 4; ++i) sum += vp[0][i][0] + vp[1][i][1];
    for (int i = 0; i < 16; ++i) sum += wf[i];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int WMODE, int PK>
void run(int wps, const float* x, float* out) {
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<WMODE, PK>), dim3(blocks), dim3(256), 0, 0, x, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<WMODE, PK>), dim3(blocks), dim3(256), 0, 0, x, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = (double)wps * iters * 8;
    const double tflops = 1024.0 * mfmas_per_simd * 4096.0 / (ms * 1e-3) / 1e12;
    static const char* names[] = {"8 x ds_read_b32 ", "4 x ds_read2_b32", "2 x ds_read_b128", "no weight reads "};
    printf("weights %s  transform %s  waves/SIMD %d : %7.1f us  %.3f of 157.3 TFLOP/s -> %.0f cycles per MFMA per SIMD at 2.4 GHz\n", names[WMODE],
           PK ? "10 pk" : "none ", wps, ms * 1e3, tflops / 157.3, ms * 1e-3 * 2.4e9 / mfmas_per_simd);
}

int main() {
    float *x, *out;
    (void)hipMalloc(&x, 1 << 20); (void)hipMalloc(&out, 64 << 20);
    (void)hipMemset(x, 0, 1 << 20);
    for (int wps = 1; wps <= 3; ++wps) {
        run<0, 1>(wps, x, out); run<1, 1>(wps, x, out); run<2, 1>(wps, x, out); run<3, 1>(wps, x, out);
        run<0, 0>(wps, x, out); run<3, 0>(wps, x, out);
    }
    return 0;
}
