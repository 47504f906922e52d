// mpc_closed_loop.h -- the receding-horizon loop AROUND the NLP solve, one instance per thread.
//
// Restates the body of CasadiOptimizer.optimize (MPC_Planner/optimizer.py:596-631) between two `sol(...)` calls:
//   * first control of the solution, forward-Euler plant step           shift_movement, optimizer.py:645-655
//   * warm start of the next solve: shifted controls / states           optimizer.py:602 with the layouts the reference
//     really produces: at step 0 the state block is the TRANSPOSED tile of the initial state, at
//     steps >= 1 the control block is the transposed (all steering rates, then all accelerations) shifted plan
//   * next parameter vector: X_ref row 0 = new state, rows 1..N = path window i+1.., frozen to the last N path points
//     once i >= L - N                                                   desired_command_and_trajectory, optimizer.py:657-702
// Row-major [B, n_w] buffers in the decision-vector order of optimizer.py:550 ([u_0..u_{N-1} | x_0..x_N]), i.e. exactly
// what mpc_solve_batch_dev consumes and produces; nx = 5 (the CasADi formulation's state) or 6 (the benchmark's extra progress
// state s, which starts at 0 and is not reported).
// Noise (`noised: True`): the reference draws from an unseeded numpy generator (optimizer.py:611-617 / 348-354), so its runs
// cannot be reproduced; here the samples come from a counter-based generator (Philox4x32-10 + Box-Muller) keyed by
// (seed; instance, step, sample index), identical on the device, in the CPU harness and in the Python mirror (noise.py):
//   mode 1  CasadiOptimizer: N(0, sigma) on the WHOLE predicted input sequence, (2, N) row-major = sample index r * N + k; the
//           noised first column is applied to the plant and the noised sequence is what gets shifted into the next warm start
//   mode 2  ForcesproOptimizer: N(0, sigma) on the applied input only (sample indices 0, 1)
// Shared by the HIP kernels (mpcgpu.hip) and the CPU emulation harness (tests/emu).
#pragma once
#include "mpc_stage_math.h"

namespace mpc {

struct LoopArgs {
    int32_t B, N, L, Lp;             // instances, horizon, closed-loop steps (iter_length), path points per instance
    const double* init_state;        // [B,5]  (x, y, delta, v, psi)            optimizer.py:575
    const double* path;              // [B,Lp,2] resampled path points          optimizer.py:691-693
    const double* orient;            // [B,Lp]   path orientation               optimizer.py:694
    const double* vdes;              // [B]      desired velocity
    double* state;                   // [B,5]  current plant state (work buffer)
    double* x0;                      // [B,n_w] warm start of the next solve
    double* p;                       // [B,n_w] parameters of the next solve
    const double* x_out;             // [B,n_w] solution of the solve just finished
    const int32_t* status;           // [B]     its per-instance status
    double* traj;                    // [B,L,5] planned states  (row i = state BEFORE step i, optimizer.py:636-638)
    double* ctrl;                    // [B,L,2] applied controls
    int32_t* step_status;            // [B,L]   solver status of every step (the reference never looks at it)
    int32_t nx;                      // 5 or 6 (rows of x0 / p / x_out have nx states per stage; traj reports the first 5)
    int32_t noise_mode;              // 0 none, 1 whole predicted input sequence (CasADi path), 2 applied input only (FORCES path)
    double sigma;                    // 0.1 lane following, 0.05 collision avoidance (optimizer.py:612-615)
    uint32_t seed_lo, seed_hi;
    const uint32_t* abort_flag;      // device word: nonzero = a solve of this loop was abandoned, the bookkeeping kernels do nothing
};

// ---- counter-based normal samples: Philox4x32-10 (Salmon et al., SC'11) + Box-Muller
MPC_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// sample j of (instance b, step i): standard normal
MPC_HD double loop_normal(const LoopArgs& A, int b, int i, int j) {
    uint32_t x[4];
    philox4x32_10((uint32_t)(j >> 1), (uint32_t)i, (uint32_t)b, 0x4D5043u, A.seed_lo, A.seed_hi, x);
    const uint64_t a1 = ((uint64_t)(x[0] >> 5) << 26) | (uint64_t)(x[1] >> 6), a2 = ((uint64_t)(x[2] >> 5) << 26) | (uint64_t)(x[3] >> 6);
    const double u1 = ((double)a1 + 1.0) * 1.1102230246251565e-16, u2 = (double)a2 * 1.1102230246251565e-16;      // 2^-53: u1 in (0,1], u2 in [0,1)
    const double r = sqrt(-2.0 * log(u1)), th = 6.283185307179586 * u2;
    return (j & 1) ? r * sin(th) : r * cos(th);
}
// noise on entry (row r, column k) of the predicted input sequence of step i (mode 1), or on the applied input (mode 2, k = 0)
MPC_HD double loop_noise(const LoopArgs& A, int b, int i, int r, int k) {
    if (A.noise_mode == 1) return A.sigma * loop_normal(A, b, i, r * A.N + k);
    if (A.noise_mode == 2 && k == 0) return A.sigma * loop_normal(A, b, i, r);
    return 0.0;
}

// reference window of step `i_next - 1` has been consumed; build X_ref for the solve of step i (optimizer.py:657-702
// is called with the index of the step that just finished)
MPC_HD void loop_write_reference(const LoopArgs& A, int b, int i_done, const double* cur) {
    const int N = A.N, nx = A.nx, nw = 2 * N + nx * (N + 1);
    double* p = A.p + (size_t)b * nw;
    for (int q = 0; q < 2 * N; ++q) p[q] = 0.0;                                  // U_ref (unused by the cost)
    double* xr = p + 2 * N;
    for (int c = 0; c < nx; ++c) xr[c] = cur[c];
    const double vd = A.vdes[b];
    for (int k = 0; k < N; ++k) {
        int idx = i_done + k + 1;
        if (i_done >= A.L - N) idx = i_done + k + 1 - (i_done - (A.L - N) + 1);    // frozen tail
        const double* pt = A.path + ((size_t)b * A.Lp + idx) * 2;
        double* r = xr + nx * (k + 1);
        r[0] = pt[0]; r[1] = pt[1]; r[2] = 0.0; r[3] = vd; r[4] = A.orient[(size_t)b * A.Lp + idx];
        if (nx == 6) r[5] = 0.0;
    }
}

// before the first solve: optimizer.py:575-594
MPC_HD void loop_setup_instance(const LoopArgs& A, int b) {
    const int N = A.N, nx = A.nx, nw = 2 * N + nx * (N + 1);
    double cur[6];
    for (int c = 0; c < nx; ++c) { cur[c] = (c < 5) ? A.init_state[(size_t)b * 5 + c] : 0.0; A.state[(size_t)b * nx + c] = cur[c]; }
    double* x0 = A.x0 + (size_t)b * nw;
    double* p = A.p + (size_t)b * nw;
    for (int q = 0; q < 2 * N; ++q) { x0[q] = 0.0; p[q] = 0.0; }
    // next_trajectories = tile(current_state) as (N+1, 5) -> p;  next_states.T.reshape(-1,1) = (5, N+1) row-major -> x0
    for (int k = 0; k <= N; ++k)
        for (int c = 0; c < nx; ++c) {
            p[2 * N + nx * k + c] = cur[c];
            x0[2 * N + c * (N + 1) + k] = cur[c];
        }
}

// after solve i: record, plant step, shift, next warm start and reference
MPC_HD void loop_advance_instance(const Params& P, const LoopArgs& A, int b, int i) {
    const int N = A.N, nx = A.nx, nw = 2 * N + nx * (N + 1);
    const double* xo = A.x_out + (size_t)b * nw;
    double cur[6], u[2], f[6], s, c, td;
    for (int q = 0; q < nx; ++q) cur[q] = A.state[(size_t)b * nx + q];
    u[0] = xo[0] + loop_noise(A, b, i, 0, 0);
    u[1] = xo[1] + loop_noise(A, b, i, 1, 0);
    for (int q = 0; q < 5; ++q) A.traj[((size_t)b * A.L + i) * 5 + q] = cur[q];
    A.ctrl[((size_t)b * A.L + i) * 2 + 0] = u[0];
    A.ctrl[((size_t)b * A.L + i) * 2 + 1] = u[1];
    if (A.step_status) A.step_status[(size_t)b * A.L + i] = A.status ? A.status[b] : 0;
    if (nx == 5) ode_eval<5>(P, cur, u, f, s, c, td); else ode_eval<6>(P, cur, u, f, s, c, td);
    for (int q = 0; q < nx; ++q) { cur[q] = cur[q] + P.dt * f[q]; A.state[(size_t)b * nx + q] = cur[q]; }
    // warm start of solve i+1.  u_end = [u_1 .. u_{N-1}, u_{N-1}] as (N,2) of the (noised) sequence; the reference flattens its
    // TRANSPOSE:
    double* x0 = A.x0 + (size_t)b * nw;
    for (int k = 0; k < N; ++k) {
        const int src = (k + 1 < N) ? k + 1 : N - 1;
        x0[k] = xo[2 * src] + (A.noise_mode == 1 ? loop_noise(A, b, i, 0, src) : 0.0);                 // all steering rates ...
        x0[N + k] = xo[2 * src + 1] + (A.noise_mode == 1 ? loop_noise(A, b, i, 1, src) : 0.0);        // ... then all accelerations
    }
    // x_f = [x_1 .. x_N, x_N] as (nx, N+1); flattened transposed = stage-major (correct order)
    for (int k = 0; k <= N; ++k) {
        const int src = (k + 1 <= N) ? k + 1 : N;
        for (int q = 0; q < nx; ++q) x0[2 * N + nx * k + q] = xo[2 * N + nx * src + q];
    }
    loop_write_reference(A, b, i, cur);
}

}  // namespace mpc
