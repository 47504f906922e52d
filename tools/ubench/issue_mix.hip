// What does a second wavefront on a SIMD buy an instruction-issue-bound wave on gfx950?
//
// The stage phases of the solver run ONE wavefront per SIMD (450+ registers per thread).  Their instruction stream is a mix: of the ~10 k
// instructions a stage thread executes per interior-point iteration ~45 % are fp64 VALU, ~20 % other VALU (moves, selects, AGPR
// spill traffic), ~30 % SALU (address arithmetic, exec-mask bookkeeping, branches) and a few per cent LDS / memory / waits
// (llvm-objdump of k_stage<6, false, 256>).  This kernel times a loop with that mix -- every instruction of one wave depends on nothing
// but its own class's previous result, so a wave is issue bound, not latency bound -- at 1, 2 and 4 wavefronts per SIMD, and the same
// loop with the work of a thread SPLIT over two wavefronts (half the instructions each).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 issue_mix.hip -o issue_mix && ./issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>

// one "unit" = 9 fp64 VALU + 4 other VALU + 6 SALU + 1 LDS  (20 instructions)
#define UNIT()                                                         \
    asm volatile(                                                      \
        "v_fma_f64 %0, %0, %4, %5\n\t"                                \
        "s_add_u32 %8, %8, 3\n\t"                                     \
        "v_fma_f64 %1, %1, %4, %5\n\t"                                \
        "v_mov_b32 %6, %7\n\t"                                        \
        "v_fma_f64 %2, %2, %4, %5\n\t"                                \
        "s_and_b32 %9, %9, 0xffff\n\t"                                \
        "v_fma_f64 %3, %3, %4, %5\n\t"                                \
        "v_add_u32 %7, %7, %6\n\t"                                    \
        "v_mul_f64 %0, %0, %4\n\t"                                    \
        "s_add_u32 %8, %8, 5\n\t"                                     \
        "v_add_f64 %1, %1, %5\n\t"                                    \
        "s_lshl_b32 %9, %9, 1\n\t"                                    \
        "v_fma_f64 %2, %2, %4, %5\n\t"                                \
        "v_mov_b32 %6, %7\n\t"                                        \
        "v_fma_f64 %3, %3, %4, %5\n\t"                                \
        "s_add_u32 %8, %8, 7\n\t"                                     \
        "v_add_f64 %0, %0, %5\n\t"                                    \
        "v_xor_b32 %7, %7, %6\n\t"                                    \
        "s_sub_u32 %9, %9, 1\n\t"                                     \
        "ds_read_b64 %10, %11\n\t"                                    \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b), "+v"(d), "+v"(m0), "+v"(m1), "+s"(s0), "+s"(s1), "=v"(l0)      \
        : "v"(laddr)                                                                                                     \
        : "scc")

template <int HALF>
__global__ void k(double* out, int units, unsigned long long* clk) {
    __shared__ double lds[1024];
    lds[threadIdx.x] = 1.0;
    __syncthreads();
    double a0 = 1.0 + threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 0.999999, d = 1e-7, l0 = 0;
    unsigned m0 = threadIdx.x, m1 = 7, s0 = 1, s1 = 3;
    const unsigned laddr = (threadIdx.x & 1023u) * 8u;
    const int n = HALF ? units / 2 : units;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        UNIT(); UNIT(); UNIT(); UNIT();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + m0 + m1 + s0 + s1 + l0;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    double* out; unsigned long long* clk;
    (void)hipMalloc(&out, 1024 * 8 * 8); (void)hipMalloc(&clk, 16 * 8);
    const int units = 500;       // x 4 x 20 = 40 000 instructions per thread
    auto report = [&](const char* what, int waves) {
        (void)hipDeviceSynchronize();
        unsigned long long c[16];
        (void)hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int w = 0; w < waves; ++w) mx = c[w] > mx ? c[w] : mx;
        printf("%-64s %8llu ticks = %.2f per instruction of the unsplit stream\n", what, mx, (double)mx / (units * 80.0));
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, units, clk);  report("1 wave / SIMD (256 threads), whole stream", 4);
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(512), 0, 0, out, units, clk);  report("2 waves / SIMD (512 threads), whole stream each (2x the work)", 8);
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(1024), 0, 0, out, units, clk); report("4 waves / SIMD (1024 threads), whole stream each (4x the work)", 16);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(512), 0, 0, out, units, clk);  report("2 waves / SIMD, HALF the stream each (the same work, split)", 8);
    }
    return 0;
}
