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
// what mpc_solve_batch_dev consumes and produces; nx = 5 (the CasADi formulation's state).
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
};

// reference window of step `i_next - 1` has been consumed; build X_ref for the solve of step i (optimizer.py:657-702
// is called with the index of the step that just finished)
MPC_HD void loop_write_reference(const LoopArgs& A, int b, int i_done, const double* cur) {
    const int N = A.N, nw = 2 * N + 5 * (N + 1);
    double* p = A.p + (size_t)b * nw;
    for (int q = 0; q < 2 * N; ++q) p[q] = 0.0;                                  // U_ref (unused by the cost)
    double* xr = p + 2 * N;
    for (int c = 0; c < 5; ++c) xr[c] = cur[c];
    const double vd = A.vdes[b];
    for (int k = 0; k < N; ++k) {
        int idx = i_done + k + 1;
        if (i_done >= A.L - N) idx = i_done + k + 1 - (i_done - (A.L - N) + 1);    // frozen tail
        const double* pt = A.path + ((size_t)b * A.Lp + idx) * 2;
        double* r = xr + 5 * (k + 1);
        r[0] = pt[0]; r[1] = pt[1]; r[2] = 0.0; r[3] = vd; r[4] = A.orient[(size_t)b * A.Lp + idx];
    }
}

// before the first solve: optimizer.py:575-594
MPC_HD void loop_setup_instance(const LoopArgs& A, int b) {
    const int N = A.N, nw = 2 * N + 5 * (N + 1);
    double cur[5];
    for (int c = 0; c < 5; ++c) { cur[c] = A.init_state[(size_t)b * 5 + c]; A.state[(size_t)b * 5 + c] = cur[c]; }
    double* x0 = A.x0 + (size_t)b * nw;
    double* p = A.p + (size_t)b * nw;
    for (int q = 0; q < 2 * N; ++q) { x0[q] = 0.0; p[q] = 0.0; }
    // next_trajectories = tile(current_state) as (N+1, 5) -> p;  next_states.T.reshape(-1,1) = (5, N+1) row-major -> x0
    for (int k = 0; k <= N; ++k)
        for (int c = 0; c < 5; ++c) {
            p[2 * N + 5 * k + c] = cur[c];
            x0[2 * N + c * (N + 1) + k] = cur[c];
        }
}

// after solve i: record, plant step, shift, next warm start and reference
MPC_HD void loop_advance_instance(const Params& P, const LoopArgs& A, int b, int i) {
    const int N = A.N, nw = 2 * N + 5 * (N + 1);
    const double* xo = A.x_out + (size_t)b * nw;
    double cur[5], u[2], f[5], s, c, td;
    for (int q = 0; q < 5; ++q) cur[q] = A.state[(size_t)b * 5 + q];
    u[0] = xo[0];
    u[1] = xo[1];
    for (int q = 0; q < 5; ++q) A.traj[((size_t)b * A.L + i) * 5 + q] = cur[q];
    A.ctrl[((size_t)b * A.L + i) * 2 + 0] = u[0];
    A.ctrl[((size_t)b * A.L + i) * 2 + 1] = u[1];
    if (A.step_status) A.step_status[(size_t)b * A.L + i] = A.status ? A.status[b] : 0;
    ode_eval<5>(P, cur, u, f, s, c, td);
    for (int q = 0; q < 5; ++q) { cur[q] = cur[q] + P.dt * f[q]; A.state[(size_t)b * 5 + q] = cur[q]; }
    // warm start of solve i+1.  u_end = [u_1 .. u_{N-1}, u_{N-1}] as (N,2); the reference flattens its TRANSPOSE:
    double* x0 = A.x0 + (size_t)b * nw;
    for (int k = 0; k < N; ++k) {
        const int src = (k + 1 < N) ? k + 1 : N - 1;
        x0[k] = xo[2 * src];                       // all steering rates ...
        x0[N + k] = xo[2 * src + 1];               // ... then all accelerations
    }
    // x_f = [x_1 .. x_N, x_N] as (5, N+1); flattened transposed = stage-major (correct order)
    for (int k = 0; k <= N; ++k) {
        const int src = (k + 1 <= N) ? k + 1 : N;
        for (int q = 0; q < 5; ++q) x0[2 * N + 5 * k + q] = xo[2 * N + 5 * src + q];
    }
    loop_write_reference(A, b, i, cur);
}

}  // namespace mpc
