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
MPC_HD void loop_advance_instance(const PRef& P, const LoopArgs& A, int b, int i) {
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

// =========================================================================================================================
// The FORCES-mode loop (ForcesproOptimizer.optimize, MPC_Planner/optimizer.py:246-366) around `solver.solve(problem)`:
//   * the guess problem["x0"] is the tiled initial point and is NEVER refreshed                              :264-274
//   * run-time parameters of step k: the next N path points / orientations (replenished with the last one), the desired
//     velocity ramping linearly to 0 over the last N steps of the run, the obstacle circle centres          :292-323
//   * the first input of the solution (+ noise on it alone when `noised`, :348-354) goes to the plant: one RK4 step   :356
// One instance per thread; buffers row-major.
// =========================================================================================================================
struct ForcesLoopArgs {
    int32_t B, N, L, Lp;             // instances, horizon, steps (iter_length = path rows the reference has), path rows per instance given
    const double* init_state;        // [B,5]
    const double* init_acc;          // [B]   initial acceleration (second entry of the guess, optimizer.py:265)
    const double* path;              // [B,Lp,2]
    const double* orient;            // [B,Lp]
    const double* vdes;              // [B]
    double obstacle[6];              // circle centres of the obstacle (problem template)
    double* state;                   // [B,5]   plant state (work buffer) = problem["xinit"] of the next solve
    double* zbar;                    // [B,N,7] problem["x0"]
    double* params;                  // [B,N,10] problem["all_parameters"]
    const double* z_out;             // [B,N,7] solution of the solve just finished
    const int32_t* exitflag;         // [B]
    double* traj;                    // [B,L,5]
    double* ctrl;                    // [B,L,2]
    int32_t* step_flag;              // [B,L] exitflag of every step (the reference asserts == 1), or null
    double dt, wheelbase;
    int32_t noise_mode;              // 0 or 2 (applied input only)
    double sigma;
    uint32_t seed_lo, seed_hi;
};

MPC_HD void forces_loop_setup_instance(const ForcesLoopArgs& A, int b) {
    double z0[7] = {0.0, A.init_acc ? A.init_acc[b] : 0.0, A.init_state[(size_t)b * 5 + 0], A.init_state[(size_t)b * 5 + 1], 0.0,
                    A.init_state[(size_t)b * 5 + 3], A.init_state[(size_t)b * 5 + 4]};
    for (int q = 0; q < 5; ++q) A.state[(size_t)b * 5 + q] = (q == 2) ? 0.0 : A.init_state[(size_t)b * 5 + q];
    for (int j = 0; j < A.N; ++j)
        for (int i = 0; i < 7; ++i) A.zbar[((size_t)b * A.N + j) * 7 + i] = z0[i];
}
// all_parameters of step k (optimizer.py:292-323)
MPC_HD void forces_loop_params_instance(const ForcesLoopArgs& A, int b, int k) {
    const int N = A.N, L = A.L;
    const double vd = A.vdes[b];
    for (int j = 0; j < N; ++j) {
        const int idx = k + 1 + j;
        const int ip = idx < A.Lp ? idx : A.Lp - 1;                                  // replenished with the last point / orientation
        // desired velocity: vdes for the first L - N steps, then linspace(vdes, 0, N); beyond the run: its last entry
        const int iv = idx < L ? idx : L - 1;
        double v = vd;
        if (iv >= L - N) {
            const int r = iv - (L - N);
            v = (N > 1) ? vd + (double)r * ((0.0 - vd) / (double)(N - 1)) : vd;      // numpy.linspace: start + i * step
            if (r == N - 1 && N > 1) v = 0.0;                                         // ... with the end point exact
        }
        double* p = A.params + ((size_t)b * N + j) * 10;
        p[0] = A.path[((size_t)b * A.Lp + ip) * 2];
        p[1] = A.path[((size_t)b * A.Lp + ip) * 2 + 1];
        p[2] = v;
        p[3] = A.orient[(size_t)b * A.Lp + ip];
        for (int q = 0; q < 6; ++q) p[4 + q] = A.obstacle[q];
    }
}
// after solve k: applied input (+ noise), record, one RK4 step of the plant
MPC_HD void forces_loop_advance_instance(const ForcesLoopArgs& A, int b, int k) {
    Params P{};
    P.dt = A.dt; P.wheelbase = A.wheelbase; P.nx = 5;
    LoopArgs nz{};
    nz.noise_mode = A.noise_mode; nz.sigma = A.sigma; nz.seed_lo = A.seed_lo; nz.seed_hi = A.seed_hi; nz.N = A.N;
    double x[5], u[2];
    for (int q = 0; q < 5; ++q) x[q] = A.state[(size_t)b * 5 + q];
    u[0] = A.z_out[((size_t)b * A.N) * 7 + 0] + loop_noise(nz, b, k, 0, 0);
    u[1] = A.z_out[((size_t)b * A.N) * 7 + 1] + loop_noise(nz, b, k, 1, 0);
    for (int q = 0; q < 5; ++q) A.traj[((size_t)b * A.L + k) * 5 + q] = x[q];
    A.ctrl[((size_t)b * A.L + k) * 2] = u[0];
    A.ctrl[((size_t)b * A.L + k) * 2 + 1] = u[1];
    if (A.step_flag) A.step_flag[(size_t)b * A.L + k] = A.exitflag ? A.exitflag[b] : 0;
    double k1[5], k2[5], k3[5], k4[5], tmp[5], s, c, td;
    const double h = A.dt;
    ode_eval<5>(P, x, u, k1, s, c, td);
    for (int i = 0; i < 5; ++i) tmp[i] = x[i] + 0.5 * h * k1[i];
    ode_eval<5>(P, tmp, u, k2, s, c, td);
    for (int i = 0; i < 5; ++i) tmp[i] = x[i] + 0.5 * h * k2[i];
    ode_eval<5>(P, tmp, u, k3, s, c, td);
    for (int i = 0; i < 5; ++i) tmp[i] = x[i] + h * k3[i];
    ode_eval<5>(P, tmp, u, k4, s, c, td);
    for (int i = 0; i < 5; ++i) A.state[(size_t)b * 5 + i] = x[i] + h / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

}  // namespace mpc
