// Which XCD does workgroup b of a launch run on?  (HIP promises nothing; the XCD-aware item orders of the convolution kernels
// assume b % 8 for 1-D grids and the linearised id % 8 for 2-D / 3-D grids.)  Prints the agreement for a few grid shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) out[lin] = xcc & 0xf;
    for (volatile int i = 0; i < 2000; ++i) {}
}
static void run(dim3 g, int threads) {
    const unsigned n = g.x * g.y * g.z;
    unsigned* d; hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, g, dim3(threads), 0, 0, d);
    std::vector<unsigned> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    unsigned ok = 0; for (unsigned i = 0; i < n; ++i) ok += (h[i] == i % 8);
    printf("grid %u,%u,%u x %d threads: %u of %u blocks on XCC (linear id %% 8); first 16:", g.x, g.y, g.z, threads, ok, n);
    for (unsigned i = 0; i < 16 && i < n; ++i) printf(" %u", h[i]);
    printf("\n"); hipFree(d);
}
int main() { run(dim3(768), 256); run(dim3(24, 8, 4), 256); run(dim3(1440), 256); run(dim3(2880), 256); run(dim3(96, 8, 1), 512); return 0; }
