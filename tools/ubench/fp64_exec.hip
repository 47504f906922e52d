// Does a wave64 VALU fp64 instruction get cheaper when only part of the EXEC mask is set?  (If the hardware skipped
// all-inactive 16-lane passes, a tile could be spread over more, emptier waves to shorten the Riccati chain.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, int active_lanes, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-9, b = 0.999999, c = 1e-7, d = 2.0 + lane * 1e-9, e = 1.000001;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (lane < active_lanes) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { a = fma(a, b, c); d = fma(d, e, c); }     // two independent dependent chains
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; unsigned long long* clk;
    hipMalloc(&out, 64 * 8 * 4); hipMalloc(&clk, 64);
    for (int al : {64, 48, 32, 16, 8, 1}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, al, 1000, clk);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, al, 1000, clk);
        hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("active lanes %2d: %.2f ticks per fp64 FMA (32000 FMAs)\n", al, (double)c / 32000.0);
    }
    return 0;
}
