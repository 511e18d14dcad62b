// Ablation probe for the conv main loop: which ingredient keeps v_mfma_f32_16x16x4_f32 from its 157 TF peak?
//   V0 registers only   V1 + B operand from LDS (ds_read_b32 at immediate offsets)   V2 + A operand from global
//   V3 = V2 + per-chunk __syncthreads + LDS re-fill (the full loop skeleton)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, int MR, int NR, int WAVES, int LDSPAD>
__global__ __launch_bounds__(WAVES * 64) void probe(const float* __restrict__ wp, const float* __restrict__ x, float* out, int nchunks) {
    constexpr int CK = 8, PS = 368, PW = 34, TAPS = 9;
    __shared__ float lds[2 * CK * PS + 2 * MR * 2 * 9 * 64 + LDSPAD];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * CK * PS + 2 * MR * 2 * 9 * 64; i += WAVES * 64) lds[i] = x[i % 1024];
    const float* asrc0 = lds + 2 * CK * PS + lane;
    __syncthreads();
    f32x4 acc[MR][NR];
    for (int m = 0; m < MR; ++m) for (int n = 0; n < NR; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
    int boff[NR];
    for (int n = 0; n < NR; ++n) { int q = ((tid >> 6) * NR + n) * 16 + (lane & 15); q %= 256; boff[n] = (q / 32) * PW + (q % 32) + (lane >> 4) * PS; }
    const float* wa = wp + lane;
    float af[2][TAPS][MR];
    for (int t = 0; t < TAPS; ++t) for (int m = 0; m < MR; ++m) { af[0][t][m] = wa[(t * MR + m) * 64]; af[1][t][m] = af[0][t][m] + 1.f; }
    float breg[NR];
    for (int n = 0; n < NR; ++n) breg[n] = lds[boff[n]];
    float sv[CK * 2];
    for (int c = 0; c < nchunks; ++c) {
        const float* bsrc = lds + (c & 1) * CK * PS;
        if (V == 3) { for (int j = 0; j < CK * 2; ++j) sv[j] = x[(c * 64 + j) * 256 % 65536 + tid]; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (V == 2 || V == 3) {
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) af[(kk + 1) & 1][t][m] = wa[(size_t)(((c * 2 + kk + 1) % 64) * TAPS * MR + t * MR + m) * 64];
            }
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                float bf[NR], afl[MR];
#pragma unroll
                for (int m = 0; m < MR; ++m) afl[m] = (V == 4) ? asrc0[(c & 1) * MR * 1152 + m * 1152 + (kk * 9 + t) * 64] : af[kk & 1][t][m];
#pragma unroll
                for (int n = 0; n < NR; ++n) bf[n] = (V >= 1) ? bsrc[boff[n] + kk * 4 * PS + (t / 3) * PW + t % 3] : breg[n];
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int n = 0; n < NR; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(afl[m], bf[n], acc[m][n], 0, 0, 0);
            }
        }
        if (V == 3) {
            float* dst = lds + ((c + 1) & 1) * CK * PS;
            for (int j = 0; j < CK * 2; ++j) dst[(j >> 1) * PS + (j & 1) * 170 + (tid % 170)] = sv[j];
            __syncthreads();
        }
    }
    float s = 0;
    for (int m = 0; m < MR; ++m) for (int n = 0; n < NR; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * WAVES * 64 + tid] = s;
}

template <int V, int MR, int NR, int WAVES, int LDSPAD = 0>
void run(const char* name, int blocks, int nchunks, const float* wp, const float* x, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<V, MR, NR, WAVES, LDSPAD>), dim3(blocks), dim3(WAVES * 64), 0, 0, wp, x, out, nchunks);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<V, MR, NR, WAVES, LDSPAD>), dim3(blocks), dim3(WAVES * 64), 0, 0, wp, x, out, nchunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double flops = (double)blocks * WAVES * nchunks * 2 * 9 * MR * NR * 2048.0;
    printf("%-28s blocks %5d chunks %2d: %8.1f us  %6.1f TFLOP/s\n", name, blocks, nchunks, ms * 1e3, flops / ms / 1e9);
}

int main() {
    float *wp, *x, *out;
    hipMalloc(&wp, 64 << 20); hipMalloc(&x, 4 << 20); hipMalloc(&out, 64 << 20);
    hipMemset(wp, 0, 64 << 20); hipMemset(x, 0, 4 << 20);
    if (getenv("PROBE_RANDOM")) {   // DVFS: zero operands clock higher than real data (MI355X_MICROARCH.md "DVFS give-back")
        std::vector<float> h(16 << 20);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2000 - 1000) * 1e-3f; }
        hipMemcpy(wp, h.data(), 64 << 20, hipMemcpyHostToDevice);
        hipMemcpy(x, h.data(), 4 << 20, hipMemcpyHostToDevice);
    }
    run<0, 2, 4, 4>("V0 regs MR2 NR4", 4096, 48, wp, x, out);
    run<4, 2, 4, 4>("V4 LDS A+B MR2 NR4", 4096, 48, wp, x, out);
    run<4, 2, 4, 4>("V4 LDS A+B MR2 NR4", 1440, 12, wp, x, out);
    run<3, 2, 4, 4>("V3 glbA+stage MR2 NR4", 4096, 48, wp, x, out);
    return 0;
}
