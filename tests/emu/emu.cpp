// tests/emu/emu.cpp -- CPU emulation harness for the HIP kernels (TEST INFRASTRUCTURE, not shipped).
//
// The build container has no GPU.  The per-thread phase functions of the kernels live in
// <package>/csrc/mpc_stage_math.h as __host__ __device__ code; this harness steps them thread by thread in
// the same order, with the same block shape, reductions and barriers as mpc_kernels.hip, so that the kernel
// math can be checked against the oracle in the `-m "not gpu"` test suite.  It is compiled into
// tests/emu/libmpc_emu.so by tests/emu/build.py and is never loaded by the product package.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../motion-planning-for-autonomous-driving-with-mpc_amd/csrc/mpc_closed_loop.h"
#include "../../motion-planning-for-autonomous-driving-with-mpc_amd/csrc/mpc_forces_qp.h"
#include "../../motion-planning-for-autonomous-driving-with-mpc_amd/csrc/mpc_host_common.h"

using namespace mpc;

template <typename R>
static void reduce_block(std::vector<R>& part, int bx, int S) {
    // part[k*bx + bl] -> every thread of instance bl gets the combination over k (k ascending, like the
    // order-independent max/min and the fixed-order sums of the GPU reduction up to rounding)
    for (int bl = 0; bl < bx; ++bl) {
        R acc = part[bl];
        for (int k = 1; k < S; ++k) red_combine(acc, part[(size_t)k * bx + bl]);
        for (int k = 0; k < S; ++k) part[(size_t)k * bx + bl] = acc;
    }
}

template <int NX>
static int run(const HostProblem& hp, int B, const double* x0, const double* p, const double* obst, double* x_out,
               int32_t* status, int32_t* iters, double* kkt, double* trace, int trace_rows, int* n_it, int bx_req) {
    const mpc_problem_desc& d = hp.desc;
    const int N = d.N, S = N + 1;
    const size_t Bp = ((size_t)B + 63) / 64 * 64;
    const int bx = bx_req > 0 ? bx_req : pick_bx(N, 512);
    const WsLayout w = ws_layout(N, NX, Bp);
    std::vector<double> ws(w.total, 0.0);
    std::vector<int32_t> iws(w.itotal, 0);
    Params P;
    fill_params(P, hp, B, Bp, bx, ws.data(), iws.data(), hp.LB.data(), hp.UB.data());
    P.x0 = x0; P.p = p; P.x_out = x_out; P.status_out = status; P.iters_out = iters; P.kkt_out = kkt;
    if (obst) {
        P.per_inst_obst = 1;
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < 6; ++i) ws[w.elem(w.OBST, i, b)] = obst[(size_t)b * 6 + i];
    }
    const int nblocks = (int)((B + bx - 1) / bx);
    const int nthreads = S * bx;
    std::vector<Ctx<NX>> ctx(nthreads);
    std::vector<Red0> r0(nthreads);
    std::vector<Red1> r1(nthreads);
    std::vector<Red2> r2(nthreads);
    std::vector<Red3> r3(nthreads);

    auto setup1 = [&](std::vector<Ctx<NX>>& cx, int blk) {
        for (int t = 0; t < nthreads; ++t) {
            Ctx<NX>& c = cx[t];
            c = Ctx<NX>{};
            c.k = t / bx;
            c.b = blk * bx + t % bx;
            c.valid = c.b < B;
            if (!c.valid) c.b = (int)Bp - 1;       // padding threads never touch memory (valid == false)
            c.active = false;
        }
    };
    auto setup = [&](int blk) { setup1(ctx, blk); };
    auto eval_finish = [&](bool reuse) {
        // neighbour-stage exchange (LDS on the GPU): x_{k+1} and lambda_{k+1} at the new iterate
        for (int t = 0; t + bx < nthreads; ++t)
            for (int i = 0; i < NX; ++i) { ctx[t].xn[i] = ctx[t + bx].z[2 + i]; ctx[t].lamn[i] = ctx[t + bx].lam[i]; }
        for (int t = 0; t < nthreads; ++t) { if (reuse) phase_eval_assemble<NX, true>(P, ctx[t], r3[t]); else phase_eval_assemble<NX, false>(P, ctx[t], r3[t]); }
        reduce_block(r3, bx, S);
        for (int t = 0; t < nthreads; ++t) phase_finish<NX>(P, ctx[t], r3[t], hp.n_mult, hp.n_z);
    };
    // ---- start-point safeguard kernel: one instance per thread
    for (int b = 0; b < B; ++b) ingest_instance<NX>(P, b);
    for (int b = 0; b < B; ++b) prestart_instance<NX>(P, b);
    // ---- init kernel
    for (int blk = 0; blk < nblocks; ++blk) {
        setup(blk);
        for (int t = 0; t < nthreads; ++t) phase_init_point<NX>(P, ctx[t], r0[t]);
        reduce_block(r0, bx, S);
        for (int t = 0; t < nthreads; ++t) phase_init_scalars<NX>(P, ctx[t], r0[t]);
        std::vector<Red3> r3i(nthreads);
        r3.swap(r3i);
        eval_finish(false);
        r3.swap(r3i);
    }
    auto record = [&](int it) {
        if (!trace || it >= trace_rows) return;
        double* tr = trace + (size_t)it * 8 * B;
        const int rows[8] = {SC_MU, SC_THETA, SC_PHI, SC_ALPHA, SC_ADU, SC_DELTA, SC_E0, SC_NTRIAL};
        for (int q = 0; q < 8; ++q)
            for (int b = 0; b < B; ++b) tr[(size_t)q * B + b] = ws[w.elem(w.SC, rows[q], b)];
    };
    int it = 0;
    const int cap = d.fixed_iters > 0 ? d.fixed_iters : d.max_iter;
    for (; it < cap; ++it) {
        int running = 0;
        for (int b = 0; b < B; ++b) running += iws[w.ielem(IS_STATUS, b)] == ST_RUNNING;
        if (!running) break;
        // ---- Riccati kernel: one instance per thread
        for (int b = 0; b < B; ++b) riccati_instance<NX>(P, b);
        // ---- stage kernel
        for (int blk = 0; blk < nblocks; ++blk) {
            setup(blk);
            bool any = false;
            for (int t = 0; t < nthreads; ++t) { PreTmp<NX> tmp; phase_load_scalars<NX>(P, ctx[t]); phase_preload<NX>(P, ctx[t], tmp); phase_premath<NX>(P, ctx[t], tmp); any |= ctx[t].active; }
            if (!any) continue;
            for (int t = 0; t < nthreads; ++t) phase_step_candidates<NX>(P, ctx[t], r1[t]);
            reduce_block(r1, bx, S);
            for (int t = 0; t < nthreads; ++t) phase_linesearch_begin<NX>(P, ctx[t], r1[t]);
            for (;;) {
                bool searching = false;
                for (int t = 0; t < nthreads; ++t) searching |= (ctx[t].active && ctx[t].searching);
                if (!searching) break;
                for (int t = 0; t < nthreads; ++t) phase_trial_eval<NX>(P, ctx[t], r2[t]);
                reduce_block(r2, bx, S);
                for (int t = 0; t < nthreads; ++t) phase_linesearch_decide<NX>(P, ctx[t], r2[t]);
            }
            for (int t = 0; t < nthreads; ++t) phase_apply_update<NX>(P, ctx[t]);
            eval_finish(true);
        }
        record(it);
    }
    record(it);
    if (n_it) *n_it = it;
    for (int b = 0; b < B; ++b) output_instance<NX>(P, b);
    return MPC_OK;
}

static int solve_any(const mpc_problem_desc* desc, const double* lbx, const double* ubx, const double* lbg,
                     const double* ubg, int32_t B, const double* x0, const double* p, const double* obst,
                     double* x_out, int32_t* status, int32_t* iters, double* kkt, double* trace,
                     int32_t trace_rows, int32_t* n_it, int32_t bx) {
    HostProblem hp;
    hp.desc = *desc;
    std::string err;
    int rc = validate_desc(hp.desc, err);
    if (rc) return rc;
    rc = set_bounds(hp, lbx, ubx, lbg, ubg, err);
    if (rc) return rc;
    if (desc->nx == 5) return run<5>(hp, B, x0, p, obst, x_out, status, iters, kkt, trace, trace_rows, n_it, bx);
    return run<6>(hp, B, x0, p, obst, x_out, status, iters, kkt, trace, trace_rows, n_it, bx);
}
extern "C" int emu_solve_batch(const mpc_problem_desc* desc, const double* lbx, const double* ubx, const double* lbg,
                               const double* ubg, int32_t B, const double* x0, const double* p, const double* obst,
                               double* x_out, int32_t* status, int32_t* iters, double* kkt, double* trace,
                               int32_t trace_rows, int32_t* n_it, int32_t bx) {
    return solve_any(desc, lbx, ubx, lbg, ubg, B, x0, p, obst, x_out, status, iters, kkt, trace, trace_rows, n_it, bx);
}
extern "C" void emu_default_desc(mpc_problem_desc* d, int32_t N, int32_t nx) { default_desc(d, N, nx); }

// closed-loop driver pieces (mpc_closed_loop.h) on host arrays: mode 0 = setup, 1 = advance after step i
extern "C" int emu_closed_loop_piece(int32_t mode, int32_t i, double dt, double wheelbase, int32_t B, int32_t N, int32_t L, int32_t Lp,
                                     const double* init_state, const double* path, const double* orient, const double* vdes,
                                     double* state, double* x0, double* p, const double* x_out, const int32_t* status,
                                     double* traj, double* ctrl, int32_t* step_status, int32_t nx, int32_t noise_mode, double sigma, uint64_t seed) {
    LoopArgs A{};
    A.B = B; A.N = N; A.L = L; A.Lp = Lp;
    A.nx = nx; A.noise_mode = noise_mode; A.sigma = sigma; A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32);
    A.init_state = init_state; A.path = path; A.orient = orient; A.vdes = vdes;
    A.state = state; A.x0 = x0; A.p = p; A.x_out = x_out; A.status = status; A.traj = traj; A.ctrl = ctrl; A.step_status = step_status;
    Params P{};
    P.dt = dt; P.wheelbase = wheelbase; P.nx = nx;
    for (int b = 0; b < B; ++b) {
        if (mode == 0) loop_setup_instance(A, b);
        else loop_advance_instance(P, A, b, i);
    }
    return 0;
}

// pieces of the FORCES-mode closed loop (mpc_closed_loop.h): mode 0 setup, 1 parameters of step k, 2 advance after solve k
extern "C" int emu_forces_loop_piece(int32_t mode, int32_t k, double dt, double wheelbase, int32_t B, int32_t N, int32_t L, int32_t Lp,
                                     const double* init_state, const double* init_acc, const double* path, const double* orient, const double* vdes,
                                     const double* obstacle, double* state, double* zbar, double* params, const double* z_out, const int32_t* exitflag,
                                     double* traj, double* ctrl, int32_t* step_flag, int32_t noise_mode, double sigma, uint64_t seed) {
    ForcesLoopArgs A{};
    A.B = B; A.N = N; A.L = L; A.Lp = Lp;
    A.init_state = init_state; A.init_acc = init_acc; A.path = path; A.orient = orient; A.vdes = vdes;
    for (int i = 0; i < 6; ++i) A.obstacle[i] = obstacle[i];
    A.state = state; A.zbar = zbar; A.params = params; A.z_out = z_out; A.exitflag = exitflag; A.traj = traj; A.ctrl = ctrl; A.step_flag = step_flag;
    A.dt = dt; A.wheelbase = wheelbase; A.noise_mode = noise_mode; A.sigma = sigma; A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32);
    for (int b = 0; b < B; ++b) {
        if (mode == 0) forces_loop_setup_instance(A, b);
        else if (mode == 1) forces_loop_params_instance(A, b, k);
        else forces_loop_advance_instance(A, b, k);
    }
    return 0;
}

// FORCES-mode SQP step (mpc_forces_qp.h), one instance after the other on host arrays
extern "C" int emu_forces_solve(int32_t B, int32_t N, double dt, double l, double wb, double rho, const double* Q, const double* R,
                                const double* Pt, const double* lb, const double* ub, const double* hl, const double* hu, int32_t hessian_mode,
                                const double* zbar, const double* params, const double* xinit, double* z_out, int32_t* iters,
                                int32_t* status, double* kkt) {
    ForcesQpArgs A{};
    A.B = B; A.Bp = B; A.N = N; A.max_it = 60;
    A.dt = dt; A.l = l; A.wb = wb; A.rho = rho; A.tol = 1e-4; A.tol_mu = 1e-6;
    for (int i = 0; i < 5; ++i) { A.Q[i] = Q[i]; A.Pt[i] = Pt[i]; }
    A.R[0] = R[0]; A.R[1] = R[1];
    forces_hessian_diag(hessian_mode, A.Q, A.R, A.Pt, A.hd, A.hdN);
    for (int i = 0; i < 7; ++i) { A.lb[i] = lb[i]; A.ub[i] = ub[i]; }
    for (int i = 0; i < 10; ++i) { A.hl[i] = hl[i]; A.hu[i] = hu[i]; }
    A.zbar = zbar; A.params = params; A.xinit = xinit; A.z_out = z_out; A.iters = iters; A.status = status; A.kkt = kkt;
    std::vector<double> ws((size_t)FQ_ROWS * N * B, 0.0);
    A.ws = ws.data();
    for (int b = 0; b < B; ++b) forces_qp_instance(A, b);
    return 0;
}
