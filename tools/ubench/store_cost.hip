// Issue cost of one wave's stores next to fp64 work (the Riccati compute wave's situation: one wave per SIMD):
//   mode 0: S coalesced 8-byte global stores (512 B per wave instruction)
//   mode 1: S/2 16-byte global stores (1 KiB per wave instruction)
//   mode 2: S 8-byte LDS writes
// plus F fp64 FMAs per iteration; one wave per workgroup, 64 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE, int S, int F>
__global__ void k(double* out, int iters, unsigned long long* clk) {
    __shared__ double lds[64 * 64];
    const int lane = threadIdx.x;
    double a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = 1.0 + lane * 1e-9 + c;
    double* base = out + (size_t)blockIdx.x * 64 * 8192;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < F; ++j) a[j & 7] = fma(a[j & 7], 0.999999, 1e-7);
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < S; ++s) base[(size_t)((i * S + s) & 4095) * 64 + lane] = a[s & 7];
        } else if (MODE == 1) {
#pragma unroll
            for (int s = 0; s < S / 2; ++s) reinterpret_cast<d2*>(base)[(size_t)((i * S + s) & 4095) * 64 + lane] = d2{a[s & 7], a[(s + 1) & 7]};
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) lds[(s & 63) * 64 + lane] = a[s & 7];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
    if (MODE == 2) out[blockIdx.x * 64 + lane] = lds[lane * 64 + (lane & 3)];
}
template <int MODE, int S, int F>
static void run(double* out, unsigned long long* clk, const char* what) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MODE, S, F>), dim3(64), dim3(64), 0, 0, out, 200, clk); (void)hipDeviceSynchronize(); }
    unsigned long long c[64]; (void)hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : c) m += (double)v; m /= 64;
    printf("%-28s x%2d, FMAs %3d per iteration: %5.0f ticks per iteration\n", what, MODE == 1 ? S / 2 : S, F, m / 200.0);
}
int main() {
    double* out; unsigned long long* clk;
    (void)hipMalloc(&out, (size_t)64 * 64 * 8192 * 8); (void)hipMalloc(&clk, 64 * 8);
    run<0, 41, 270>(out, clk, "global 8 B stores"); run<0, 41, 0>(out, clk, "global 8 B stores"); run<0, 8, 270>(out, clk, "global 8 B stores");
    run<1, 42, 270>(out, clk, "global 16 B stores"); run<1, 42, 0>(out, clk, "global 16 B stores");
    run<2, 41, 270>(out, clk, "LDS 8 B writes"); run<2, 41, 0>(out, clk, "LDS 8 B writes");
    return 0;
}
