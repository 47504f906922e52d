// fp64 issue cost of one wave64 on gfx950:
//   (1) does a partially filled EXEC mask make a VALU fp64 instruction cheaper?  (no)
//   (2) dependent-issue latency vs throughput: C independent FMA chains interleaved, C = 1, 2, 4, 8
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C>
__global__ void k(double* out, int active_lanes, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    double a[C];
#pragma unroll
    for (int c = 0; c < C; ++c) a[c] = 1.0 + lane * 1e-9 + c;
    const double b = 0.999999, d = 1e-7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (lane < active_lanes) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32 / C; ++j) {
#pragma unroll
                for (int c = 0; c < C; ++c) a[c] = fma(a[c], b, d);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int C>
static void run(double* out, unsigned long long* clk, int al) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<C>, dim3(1), dim3(64), 0, 0, out, al, 1000, clk); (void)hipDeviceSynchronize(); }
    unsigned long long c;
    (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("chains %d, active lanes %2d: %.2f ticks per fp64 FMA\n", C, al, (double)c / 32000.0);
}
int main() {
    double* out; unsigned long long* clk;
    (void)hipMalloc(&out, 64 * 8 * 4); (void)hipMalloc(&clk, 64);
    for (int al : {64, 32, 16, 1}) run<2>(out, clk, al);
    run<1>(out, clk, 64); run<2>(out, clk, 64); run<4>(out, clk, 64); run<8>(out, clk, 64);
    return 0;
}
