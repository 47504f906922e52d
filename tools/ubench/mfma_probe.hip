// Probe of the cross-lane machinery the wave-per-instance Riccati (csrc/mpc_riccati_mfma.h) is built from, on the GPU it runs on:
//   1. the lane -> element maps of v_mfma_f64_4x4x4_f64 (4 blocks of 4x4x4; the guide documents the 16x16x4 form only): for every
//      lane of A a one-hot A against B[l] = l + 1 shows which D lanes it feeds and which B lane it is paired with;
//   2. dependent-chain latencies (shader-clock ticks per instruction of a chain of 256): the MFMA itself, ds_bpermute, ds_swizzle,
//      a DPP row shift, v_permlane32_swap, v_readlane feeding a VALU instruction, fp64 FMA, fp64 division.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void __launch_bounds__(64) k_map(double* out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        const double a = (lane == la) ? 1.0 : 0.0, b = (double)(lane + 1);
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[la * 64 + lane] = d;
    }
}

#define SWZ(v, pat) v = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), pat), __builtin_amdgcn_ds_swizzle(__double2loint(v), pat))
__device__ __forceinline__ double bperm(double v, int addr) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ double rdlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <int MODE>
__global__ void __launch_bounds__(64) k_lat(double* out, unsigned long long* clk, int reps) {
    const int lane = threadIdx.x;
    double x = 1.0 + 1e-3 * lane, y = 0.5 + 1e-4 * lane;
    const int addr = ((lane + 17) & 63) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MODE == 0) x = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, x, 0, 0, 0);
            if (MODE == 1) x = bperm(x, addr);
            if (MODE == 2) SWZ(x, 0x401F);
            if (MODE == 3) x = dpp<0x104>(x) + y;                 // row_shl:4, then one fp64 add so that the chain is VALU -> DPP -> VALU
            if (MODE == 4) {
                int hi = __double2hiint(x), lo = __double2loint(x), hi2 = hi, lo2 = lo;
                auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi2, false, false);
                auto r2 = __builtin_amdgcn_permlane32_swap(lo, lo2, false, false);
                x = __hiloint2double(r1[0], r2[0]) + y;
            }
            if (MODE == 5) x = fma(rdlane(x, 10), y, y);
            if (MODE == 6) x = fma(x, y, y);
            if (MODE == 7) x = y / x + 1.0;
            if (MODE == 8) x = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, (__attribute__((ext_vector_type(4))) double){x, x, x, x}, 0, 0, 0)[0];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = x;
    if (lane == 0) clk[0] = t1 - t0;
}

template <int MODE>
static double lat(double* out, unsigned long long* clk) {
    const int reps = 64;
    for (int i = 0; i < 2; ++i) { hipLaunchKernelGGL(k_lat<MODE>, dim3(1), dim3(64), 0, 0, out, clk, reps); (void)hipDeviceSynchronize(); }
    unsigned long long c;
    (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    return (double)c / (reps * 16);
}

int main() {
    double* out; unsigned long long* clk;
    (void)hipMalloc(&out, 64 * 64 * 8); (void)hipMalloc(&clk, 64);
    hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, out);
    (void)hipDeviceSynchronize();
    static double h[64 * 64];
    (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("v_mfma_f64_4x4x4_f64: A lane -> (D lane : paired B lane) ...\n");
    for (int la = 0; la < 64; ++la) {
        printf("A%02d:", la);
        for (int d = 0; d < 64; ++d) if (h[la * 64 + d] != 0.0) printf(" D%02d:B%02d", d, (int)h[la * 64 + d] - 1);
        printf("\n");
    }
    printf("dependent-chain ticks per instruction (s_memtime):\n");
    printf("  mfma_f64_4x4x4 %.1f\n  ds_bpermute (double = 2) %.1f\n  ds_swizzle (double = 2) %.1f\n  dpp row_shl + add %.1f\n  permlane32_swap (2) + add %.1f\n"
           "  readlane (2) + fma %.1f\n  fma f64 %.1f\n  div f64 + add %.1f\n  mfma_f64_16x16x4 %.1f\n",
           lat<0>(out, clk), lat<1>(out, clk), lat<2>(out, clk), lat<3>(out, clk), lat<4>(out, clk), lat<5>(out, clk), lat<6>(out, clk), lat<7>(out, clk), lat<8>(out, clk));
    return 0;
}
