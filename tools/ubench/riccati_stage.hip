// One backward Riccati stage of the nx = 6 recursion, three ways -- the evidence behind "MFMA is not used" (DESIGN.md section 4):
//   scalar   the product's own step (riccati_backward_step of csrc/mpc_stage_math.h: sparse A = I + dt F, one instance per lane,
//            ~290 fp64 VALU instructions per stage), operands in registers, no memory traffic; also its matrix half alone
//   dense    the same recursion with A, B treated as dense 6x6 / 6x2 blocks on the VALU, one instance per lane (what a
//            structure-blind formulation costs)
//   mfma     the instruction mix of a v_mfma_f64_4x4x4 formulation, ONE instance per wave: P and A padded to 8x8 = 2x2 blocks of
//            4x4, P A and A' (P A) are 2 MFMA instructions each (4 block products per instruction, two k-steps), G' K one more;
//            between dependent products the 8x8 result has to change from the D layout to the A/B operand layout (modelled by 16
//            ds_swizzle per stage).  Numerically meaningless -- it times the dependent instruction chain a real kernel would issue.
// Reports shader-clock ticks per stage of one wave and per (instance, stage).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../motion-planning-for-autonomous-driving-with-mpc_amd/csrc/mpc_stage_math.h"
using namespace mpc;

template <int MODE>      // 0: full scalar step, 1: matrix half only
__global__ void __launch_bounds__(64) k_scalar(double* out, int stages, unsigned long long* clk) {
    constexpr int NX = 6;
    using D = Dim<NX>;
    const int lane = threadIdx.x;
    Params P0{};
    P0.dt = 0.1;
    const PRef P(P0);
    RicStage<NX> s;
    for (int i = 0; i < D::NS; ++i) s.H[i] = 0.0;
    for (int i = 0; i < NX; ++i) { s.H[D::sidx(i, i)] = 2.0 + 0.01 * lane + i; s.gx[i] = 0.1 * i; s.cn[i] = 1e-3 * (i + lane); }
    s.ruu[0] = 4.0; s.ruu[1] = 0.4; s.gu[0] = 0.01; s.gu[1] = 0.02;
    for (int i = 0; i < 6; ++i) s.a[i] = 0.01 * (i + 1) + 1e-4 * lane;
    double Ps[D::NS], pv[NX];
    for (int i = 0; i < D::NS; ++i) Ps[i] = s.H[i];
    for (int i = 0; i < NX; ++i) pv[i] = s.gx[i];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < stages; ++k) {
        double Pn[D::NS];
#pragma unroll
        for (int i = 0; i < D::NS; ++i) Pn[i] = Ps[i];
        RicGain<NX> g;
        ric_matrix_step<NX>(P, 1, s, 0.0, 0.0, 0.0, Ps, g);
        if (MODE == 0) ric_vector_step<NX>(P, s, Pn, g, pv);
#pragma unroll
        for (int i = 0; i < D::NS; ++i) Ps[i] = Ps[i] * 0.5 + s.H[i] * 0.5;        // keep the recursion bounded
        s.a[0] += g.K0[0] * 1e-12;                                                  // keep the gains alive
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double acc = 0.0;
    for (int i = 0; i < D::NS; ++i) acc += Ps[i];
    for (int i = 0; i < NX; ++i) acc += pv[i];
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(64) k_dense(double* out, int stages, unsigned long long* clk) {
    constexpr int n = 6, m = 2;
    const int lane = threadIdx.x;
    double A[n][n], B[n][m], Pm[n][n], H[n][n], R[m][m];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { A[i][j] = (i == j) + 0.01 * (i + j) + 1e-4 * lane; H[i][j] = (i == j) * 2.0; Pm[i][j] = H[i][j]; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) B[i][j] = 0.1 * (i == 2 + j);
    R[0][0] = 4.0; R[1][1] = 0.4; R[0][1] = R[1][0] = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < stages; ++k) {
        double PA[n][n], PB[n][m], G[m][n], L[m][m], K[m][n], Pn[n][n];
#pragma unroll
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < n; ++j) { double t = 0; for (int r = 0; r < n; ++r) t += Pm[i][r] * A[r][j]; PA[i][j] = t; }
#pragma unroll
            for (int j = 0; j < m; ++j) { double t = 0; for (int r = 0; r < n; ++r) t += Pm[i][r] * B[r][j]; PB[i][j] = t; }
        }
#pragma unroll
        for (int i = 0; i < m; ++i) {
#pragma unroll
            for (int j = 0; j < n; ++j) { double t = 0; for (int r = 0; r < n; ++r) t += B[r][i] * PA[r][j]; G[i][j] = t; }
#pragma unroll
            for (int j = 0; j < m; ++j) { double t = R[i][j]; for (int r = 0; r < n; ++r) t += B[r][i] * PB[r][j]; L[i][j] = t; }
        }
        const double idet = 1.0 / (L[0][0] * L[1][1] - L[0][1] * L[1][0]);
#pragma unroll
        for (int j = 0; j < n; ++j) {
            K[0][j] = -(L[1][1] * G[0][j] - L[0][1] * G[1][j]) * idet;
            K[1][j] = -(-L[1][0] * G[0][j] + L[0][0] * G[1][j]) * idet;
        }
#pragma unroll
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = i; j < n; ++j) {
                double t = H[i][j] + G[0][i] * K[0][j] + G[1][i] * K[1][j];
                for (int r = 0; r < n; ++r) t += A[r][i] * PA[r][j];
                Pn[i][j] = Pn[j][i] = t;
            }
#pragma unroll
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = 0; j < n; ++j) Pm[i][j] = 0.5 * Pn[i][j] + 0.5 * H[i][j];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double acc = 0.0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) acc += Pm[i][j];
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(64) k_mfma(double* out, int stages, unsigned long long* clk) {
    const int lane = threadIdx.x;
    double p = 1.0 + 1e-3 * lane, a = 0.5 + 1e-4 * lane, h = 2.0, acc0 = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < stages; ++k) {
        // P A : two k-steps over the four 4x4 output blocks
        double pa = __builtin_amdgcn_mfma_f64_4x4x4f64(p, a, 0.0, 0, 0, 0);
        pa = __builtin_amdgcn_mfma_f64_4x4x4f64(p, a, pa, 0, 0, 0);
        // D layout -> operand layout (4 swizzles of two dwords each)
        double pb = pa;
#define SWZ(v, pat) v = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), pat), __builtin_amdgcn_ds_swizzle(__double2loint(v), pat))
        SWZ(pb, 0x041F); SWZ(pb, 0x081F); SWZ(pb, 0x101F); SWZ(pb, 0x201F);
        // A' (P A)
        double apa = __builtin_amdgcn_mfma_f64_4x4x4f64(a, pb, h, 0, 0, 0);
        apa = __builtin_amdgcn_mfma_f64_4x4x4f64(a, pb, apa, 0, 0, 0);
        // G = B'(P A) rows, Lam^-1 (2x2, VALU on broadcast entries), K: ~12 VALU fp64
        double g = pa * 0.1, idet = 1.0 / (fma(g, g, 4.0));
        double kk = -g * idet;
        SWZ(kk, 0x041F); SWZ(kk, 0x081F); SWZ(kk, 0x101F); SWZ(kk, 0x201F);
        // + G' K (rank 2, padded to one 4x4x4 step)
        p = __builtin_amdgcn_mfma_f64_4x4x4f64(g, kk, apa, 0, 0, 0);
        p = 0.5 * p + 1.0;
        acc0 += p;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc0 + p;
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

template <class K>
static double run(K kern, double* out, unsigned long long* clk, int stages) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(64), dim3(64), 0, 0, out, stages, clk); (void)hipDeviceSynchronize(); }
    unsigned long long c[64];
    (void)hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : c) m += (double)v;
    return m / 64 / stages;
}
int main() {
    double* out; unsigned long long* clk;
    (void)hipMalloc(&out, 64 * 64 * 8); (void)hipMalloc(&clk, 64 * 8);
    const int stages = 3100;
    const double full = run(k_scalar<0>, out, clk, stages), mat = run(k_scalar<1>, out, clk, stages), dense = run(k_dense, out, clk, stages), mf = run(k_mfma, out, clk, stages);
    printf("per stage of one wave (s_memtime ticks)        per (instance, stage)\n");
    printf("scalar sparse step, 64 instances per wave: %7.0f   %7.1f\n", full, full / 64);
    printf("  its matrix half alone:                   %7.0f   %7.1f\n", mat, mat / 64);
    printf("dense 6x6 blocks on the VALU, 64 per wave: %7.0f   %7.1f\n", dense, dense / 64);
    printf("MFMA f64 4x4x4 mix, 1 instance per wave:   %7.0f   %7.1f\n", mf, mf);
    return 0;
}
