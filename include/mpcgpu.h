/*
 * mpcgpu.h -- C-ABI of the MI355X-native batched NLP solver for the receding-horizon optimisation of
 *             TGoldC/Motion-Planning-for-Autonomous-Driving-with-MPC (MPC_Planner/optimizer.py).
 *
 * This is the drop-in boundary.  What each entry point replaces in the reference:
 *
 *   mpc_create / mpc_destroy ........ `ca.nlpsol('solver','ipopt', nlp_prob, opts_setting)`
 *                                     (MPC_Planner/optimizer.py:513-560, rebuilt every step at :605) and
 *                                     `model.generate_solver(options=codeoptions)` (optimizer.py:197-245).
 *                                     The handle owns the device workspace; create it once, reuse it.
 *   mpc_set_bounds .................. the `lbg, ubg, lbx, ubx` lists of `inequal_constraints()`
 *                                     (optimizer.py:413-491) that the reference passes to every `sol(...)` call.
 *   mpc_solve_batch ................. `res = sol(x0=init_control, p=c_p, lbg=lbg, lbx=lbx, ubg=ubg, ubx=ubx)`
 *                                     (optimizer.py:607) for B independent instances at once; row b of x_out is
 *                                     `res['x'].full().ravel()` of instance b (optimizer.py:609).
 *                                     FORCESPRO twin: `solver.solve(problem)` (optimizer.py:326), C side
 *                                     `FORCESNLPsolver_solve(params, output, info, fs, extfunc)`
 *                                     (test/FORCESNLPsolver/include/FORCESNLPsolver.h:219); the status codes
 *                                     below follow that header's exit codes (FORCESNLPsolver.h:68-106).
 *   mpc_solve_batch_dev ............. same with device-resident buffers on a caller-supplied HIP stream
 *                                     (no PCIe traffic; this is what bench.py times).
 *   mpc_plant_step .................. `shift_movement` plant update `x0 + delta_t * f(x0, u[:,0])`
 *                                     (optimizer.py:645-650) / `model.eq` RK4 step (optimizer.py:98,356).
 *   mpc_closed_loop_batch[_dev] ..... the loop body of `CasadiOptimizer.optimize` between two solves (optimizer.py:596-631,
 *                                     645-702): first control, plant step, shifted warm start, next reference window.
 *   mpc_forces_stage_eval ........... `FORCESNLPsolver_casadi2forces` (test/FORCESNLPsolver/FORCESNLPsolver_interface.c:41-198):
 *                                     FORCES-mode stage cost / RK4 dynamics / inequalities with their derivatives.
 *   mpc_forces_solve_batch .......... `output, exitflag, info = solver.solve(problem)` of ForcesproOptimizer (optimizer.py:326).
 *   mpc_forces_closed_loop_batch .... the loop of ForcesproOptimizer.optimize around it (optimizer.py:246-366).
 *   mpc_metrics_batch ............... deviation.txt / RMSD.txt of MPCPlanner (mpc_planner.py:184-199, 279-292), circle clearance.
 *   mpc_validity_batch .............. the collision / road-boundary verdict of test/test_mpc_planner.py:37-47.
 *
 * Conventions (modelled on FORCESNLPsolver.h:117-203): caller-owned plain buffers, int return codes, no
 * exceptions, no callbacks, no globals; the library never keeps a host pointer past the call.
 * Row layout of x0 / p / x_out is the reference's decision-vector order (optimizer.py:550,552):
 *     [u_0(2) u_1(2) ... u_{N-1}(2) | x_0(nx) x_1(nx) ... x_N(nx)],   n_w = 2 N + nx (N + 1)
 * All floating point is IEEE double.  A handle is not thread-safe: one handle per host thread / stream.
 */
#ifndef MPCGPU_H
#define MPCGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCGPU_ABI_VERSION 1

/* per-instance numerical status (FORCESNLPsolver.h:68-106 semantics) */
#define MPC_STATUS_CONVERGED   1    /* OPTIMAL: scaled KKT error <= tol                        */
#define MPC_STATUS_MAXITER     0    /* MAXITREACHED                                            */
#define MPC_STATUS_NAN        (-6)  /* BADFUNCEVAL: NaN/Inf met in a function evaluation       */
#define MPC_STATUS_NOPROGRESS (-7)  /* NOPROGRESS: line search or regularisation gave up       */

/* API return codes (0 = ok, negative = error; mpc_last_error() has the text) */
#define MPC_OK                 0
#define MPC_ERR_INVALID      (-1)   /* bad argument / unsupported problem shape                */
#define MPC_ERR_HIP          (-2)   /* HIP runtime failure (no device, OOM, launch error)      */
#define MPC_ERR_BOUNDS       (-3)   /* lbg/ubg do not have the structure of optimizer.py:421-469 */
#define MPC_ERR_STATE        (-4)   /* call order (e.g. solve before set_bounds)               */

/* formulation of the stage model */
#define MPC_FORM_CASADI_EULER  0    /* CasadiOptimizer: forward Euler multiple shooting (optimizer.py:380-382) */

typedef struct mpc_handle mpc_handle;

typedef struct mpc_problem_desc {
    int32_t N;             /* predict_horizon (optimizer.py:520); 1 <= N <= 127                          */
    int32_t nx;            /* 5 = reference state [x, y, delta, v, psi]; 6 appends a decoupled progress
                              state s (s' = v, no weight, unbounded) used by the synthetic benchmark      */
    int32_t nu;            /* must be 2: [deltaDot, aLong]                                                */
    int32_t formulation;   /* MPC_FORM_CASADI_EULER                                                       */
    int32_t max_iter;      /* ipopt.max_iter = 100 (optimizer.py:556)                                     */
    int32_t fixed_iters;   /* 0 = iterate to convergence; k > 0 = exactly k iterations (benchmark mode)   */
    int32_t obst_mult;     /* 3: each of the 3 circle-pair distances is appended 3x (optimizer.py:395-403)*/
    int32_t device;        /* HIP device ordinal                                                          */
    double dt;             /* scenario.dt (optimizer.py:52)                                               */
    double wheelbase;      /* p.a + p.b = 2.5789128 (configuration.py:362-363)                            */
    double friction_div;   /* literal 2.578 of optimizer.py:378                                           */
    double ego_offset;     /* centre offset of the front/rear ego circle, 0.75 (configuration.py:69-93)   */
    double tol;            /* IPOPT tol, 1e-8                                                             */
    double Q[8];           /* diag(weight_x, weight_y, weight_steering_angle, weight_velocity,
                              weight_heading_angle[, 0]) (optimizer.py:500-502)                           */
    double R[2];           /* diag(weight_velocity_steering_angle, weight_long_acceleration) (:503)       */
    double P[8];           /* terminal weights (:504-505) -- built but unused by the CasADi cost (dead
                              expression at :510); kept for the FORCES formulation                        */
    double obstacle[6];    /* obstacle circle centres (x,y) x 3: centre, front, rear (optimizer.py:60-64) */
} mpc_problem_desc;

/* fills *desc with the reference defaults for horizon N and nx in {5,6} (ZAM_Over-1_1 lane-following
 * weights, dummy obstacle at (-100,0)) */
void mpc_default_desc(mpc_problem_desc* desc, int32_t N, int32_t nx);

int mpc_create(mpc_handle** out, const mpc_problem_desc* desc);
int mpc_destroy(mpc_handle* h);
const char* mpc_last_error(const mpc_handle* h);   /* h may be NULL: error of the last failed mpc_create */

/* lbx/ubx: n_w entries (optimizer.py:470-491), +-inf = absent.  lbg/ubg: n_g = 1 + nx(N+1) + 9(N+1) entries
 * (optimizer.py:421-469): row 0 friction [lo, hi]; nx(N+1) equality rows (lbg == ubg); 9(N+1) obstacle rows,
 * all with the same [lo, hi].  Anything else -> MPC_ERR_BOUNDS.  A friction lower bound <= 0 is implied by the
 * absolute value in the row and gets no barrier.  Passing NULL for all four installs the reference defaults. */
int mpc_set_bounds(mpc_handle* h, const double* lbx, const double* ubx, const double* lbg, const double* ubg);

/* Host-buffer entry point.  x0, p, x_out: [B, n_w] row-major.  obst: [B, 6] per-instance obstacle circle
 * centres or NULL (shared centres of the descriptor).  status/iters/kkt: [B], any may be NULL.            */
int mpc_solve_batch(mpc_handle* h, int32_t B, const double* x0, const double* p, const double* obst,
                    double* x_out, int32_t* status, int32_t* iters, double* kkt);

/* Second chance (both entry points, on unless the option "rescue" is "0"; not in fixed_iters mode): instances whose line search
 * or regularisation gave up (status -7, where IPOPT would enter its feasibility-restoration phase) or that ran out of
 * iterations are re-solved on the device with the lower bound of the circle-distance rows raised in steps from 0 to its value,
 * each level warm-started from the last; the final level is the original problem, so a row that comes back with status 1 is a
 * KKT point of the ORIGINAL NLP to the original tolerance (possibly another local optimum than IPOPT's); rows that still fail keep
 * their first result and status.  iters[] includes the iterations of the second chance.  mpc_last_rescued: how many instances
 * of the last solve took it.                                                                                              */
int mpc_last_rescued(const mpc_handle* h);

/* Device-buffer entry point: same arguments as device pointers, work enqueued on `stream` (a hipStream_t,
 * NULL = default stream).  Returns after enqueueing + the convergence polls; the outputs are complete when
 * the stream is synchronised (the call itself synchronises the stream unless fixed_iters > 0).            */
int mpc_solve_batch_dev(mpc_handle* h, int32_t B, const double* d_x0, const double* d_p, const double* d_obst,
                        double* d_x_out, int32_t* d_status, int32_t* d_iters, double* d_kkt, void* stream);

/* Batched plant step on the device path: x_next = x + dt f(x,u) (integrator 0 = forward Euler,
 * optimizer.py:649-650) or one RK4 step (integrator 1, optimizer.py:97-98).  x: [B, nx], u: [B, 2] host. */
int mpc_plant_step(mpc_handle* h, int32_t B, int32_t integrator, const double* x, const double* u, double* x_next);
/* same with device pointers; the kernel is enqueued on `stream`, nothing is synchronised or allocated */
int mpc_plant_step_dev(mpc_handle* h, int32_t B, int32_t integrator, const double* d_x, const double* d_u, double* d_x_next, void* stream);

/* Closed-loop driver (scope row f1): the loop body of CasadiOptimizer.optimize (optimizer.py:596-631) run for B egos
 * without host round trips between the solves -- per step: mpc_solve_batch_dev, first control, forward-Euler plant
 * step (shift_movement, optimizer.py:645-655), shifted warm start in the layouts the reference produces
 * (optimizer.py:602), next reference window incl. the frozen tail (desired_command_and_trajectory, :657-702).
 * init_state [B,5] = (x, y, 0, v, psi) (optimizer.py:575); path [B,Lp,2], orient [B,Lp]: resampled path
 * points and orientation of every ego, Lp >= L = iter_length >= N; vdes [B].  Outputs: traj [B,L,5] (row i = state
 * before step i: what optimize() returns as its first array), ctrl [B,L,2] (second array), step_status [B,L] or NULL
 * (solver status of every step; the reference ignores it).  No noise (`noised: False`).                           */
int mpc_closed_loop_batch(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* init_state, const double* path,
                          const double* orient, const double* vdes, double* traj, double* ctrl, int32_t* step_status);
/* The same loop with the reference's `noised: True` behaviour, reproducibly: N(0, sigma) samples from a counter-based generator
 * (Philox4x32-10 + Box-Muller, keyed by seed and (instance, step, sample); csrc/mpc_closed_loop.h, mirrored by noise.py).
 * noise_mode 0: none; 1: CasadiOptimizer (optimizer.py:611-617) -- noise on the whole predicted input sequence, the noised first
 * column is applied and the noised sequence is shifted into the next warm start; 2: ForcesproOptimizer (optimizer.py:348-354) --
 * noise on the applied input only.  sigma: 0.1 lane following, 0.05 collision avoidance (the reference's values).
 * nx = 5 or 6 (the progress state of nx = 6 starts at 0 and is not reported: traj stays [B,L,5]).
 * The device-pointer form enqueues the WHOLE loop without a host synchronisation per step when the solves run in the persistent
 * pipeline launch; an abandoned launch or an instance that needs the second chance is noticed once, at the end, and the loop is
 * then replayed step by step (mpc_last_loop_replayed tells).  Option "loop_async" = "0" forces the step-by-step form.          */
int mpc_closed_loop_batch_ex(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* init_state, const double* path,
                             const double* orient, const double* vdes, int32_t noise_mode, double sigma, uint64_t seed,
                             double* traj, double* ctrl, int32_t* step_status);
int mpc_closed_loop_batch_dev_ex(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* d_init_state, const double* d_path,
                                 const double* d_orient, const double* d_vdes, int32_t noise_mode, double sigma, uint64_t seed,
                                 double* d_traj, double* d_ctrl, int32_t* d_step_status, void* stream);
int mpc_last_loop_replayed(const mpc_handle* h);
/* FORCES-mode stage functions (scope row a11): what `FORCESNLPsolver_casadi2forces` evaluates per stage
 * (test/FORCESNLPsolver/FORCESNLPsolver_interface.c:41-198 -> casadi_f0..f9, FORCESNLPsolver_model.c:75-1756; the model
 * of ForcesproOptimizer, optimizer.py:91-245) for B independent (z, p) pairs.  z [B,7] = (deltaDot, aLong, x, y, delta, v,
 * psi), p [B,10] = (x_ref, y_ref, v_des, psi_ref, obstacle centre / front / rear circle).  Outputs, row-major, any may be
 * NULL: f [B], grad_f [B,7], c [B,5] (one RK4 step of length dt), jac_c [B,5,7], h [B,10] (friction circle, nine squared
 * circle distances), jac_h [B,10,7].  terminal != 0: the last stage (terminal weights P, no input cost, no c).
 * Weights: Q / R / P of the handle's descriptor; dt, wheelbase (ODE), friction_div (2.578 in psi_dot), ego_offset too. */
int mpc_forces_stage_eval(mpc_handle* h, int32_t B, int32_t terminal, const double* z, const double* p, double* f,
                          double* grad_f, double* c, double* jac_c, double* hval, double* jac_h);

/* FORCES-mode solve (scope row f3): `output, exitflag, info = solver.solve(problem)` of ForcesproOptimizer
 * (optimizer.py:326; C side FORCESNLPsolver_solve, test/FORCESNLPsolver/include/FORCESNLPsolver.h:117-219) for B independent
 * problems: ONE step of sequential quadratic programming from the guess x0, as the reference configures FORCESPRO
 * (sqp_nlp.maxqps = 1, BFGS initialised to 2.5 I, reg_hessian 5e-6; optimizer.py:225-240).  Horizon N = the descriptor's
 * N, weights Q / R / P of the descriptor.  x0 [B,N,7] = problem["x0"], xinit [B,5], all_parameters [B,N,10] (optimizer.py:
 * 124-127, 313-318); lb/ub [7], hl/hu [10] = inequal_constraint() (optimizer.py:100-119; +-inf or |v| >= 1e300 = absent).
 * hessian_mode 0: QP Hessian = exact Hessian of the least-squares cost (Gauss-Newton SQP, default); 1: the literal
 * `bfgs_init = 2.5 I` (see csrc/mpc_forces_qp.h: forces_hessian_diag for why that is not the default).
 * Outputs: x_out [B,N,7] = output["x01".."xN"]; exitflag [B] (1 solved, 0 iteration limit, -6 NaN, -7 inconsistent
 * linearised constraints; FORCESNLPsolver.h:68-106); it [B] interior-point iterations; res [B] final residual.          */
int mpc_forces_solve_batch(mpc_handle* h, int32_t B, const double* x0, const double* xinit, const double* all_parameters,
                           const double* lb, const double* ub, const double* hl, const double* hu, int32_t hessian_mode,
                           double* x_out, int32_t* exitflag, int32_t* it, double* res);

/* The FORCES-mode closed loop around that solve (ForcesproOptimizer.optimize, optimizer.py:246-366) for B egos, on the device with
 * nothing coming back to the host between the steps: the never-refreshed guess problem["x0"] = tiled initial point (:264-274), the
 * run-time parameters of every step (next N path points / orientations replenished with the last one, desired velocity ramping to 0
 * over the last N steps of the run, obstacle circle centres of the descriptor; :292-323), one SQP step, first input (+ N(0, sigma)
 * on it alone with noise_mode 2, :348-354; 0 = none; seeded counter-based samples as in mpc_closed_loop_batch_ex), one RK4 plant step
 * (:356).  init_state [B,5], init_acc [B] or NULL (0), path [B,Lp,2], orient [B,Lp] (the reference has Lp = L = iter_length), vdes [B];
 * lb/ub/hl/hu as in mpc_forces_solve_batch (host arrays).  Outputs traj [B,L,5], ctrl [B,L,2], step_flag [B,L] or NULL (the exitflag of
 * every step; the reference asserts it is 1, optimizer.py:330).                                                                  */
int mpc_forces_closed_loop_batch(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* init_state, const double* init_acc,
                                 const double* path, const double* orient, const double* vdes, const double* lb, const double* ub,
                                 const double* hl, const double* hu, int32_t hessian_mode, int32_t noise_mode, double sigma, uint64_t seed,
                                 double* traj, double* ctrl, int32_t* step_flag);
int mpc_forces_closed_loop_batch_dev(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* d_init_state, const double* d_init_acc,
                                     const double* d_path, const double* d_orient, const double* d_vdes, const double* lb, const double* ub,
                                     const double* hl, const double* hu, int32_t hessian_mode, int32_t noise_mode, double sigma, uint64_t seed,
                                     double* d_traj, double* d_ctrl, int32_t* d_step_flag, void* stream);

/* Post-hoc trajectory metrics (scope row f4) for B planned trajectories traj [B,L,5] (host buffers, any output may be NULL):
 *   deviation [B,L]  distance to the nearest point of origin_path [B,Lo,2]   (plot_deviation_euclidean_dis,
 *                    mpc_planner.py:184-199; find_closest_point, configuration.py:26-37)
 *   rmsd [B,2]       sqrt(sum_i (ref_path[i] - x[i])^2 / (L-1)) of x and y   (compute_rmsd, mpc_planner.py:279-292)
 *   clearance [B]    min over steps and circle pairs of distance - r_sum; all_pairs = 0: the three pairs (ego circle j,
 *                    obstacle circle j) that optimizer.py:395-403 constrains (each three times), all_pairs = 1: all nine
 *                    pairs; circle centres of the handle's problem template                                          */
int mpc_metrics_batch(mpc_handle* h, int32_t B, int32_t L, int32_t Lo, const double* traj, const double* ref_path,
                      const double* origin_path, double r_sum, int32_t all_pairs, double* deviation, double* rmsd,
                      double* clearance);
/* Collision / road verdict of B planned trajectories traj [B,L,5] (scope row f4): what the reference's test asks of
 * commonroad_dc's collision checker (test/test_mpc_planner.py:37-47) -- does the ego rectangle (mpc_planner.py:99: length 4.3,
 * width 1.8, centred on the planned position, heading psi) of some step overlap an obstacle rectangle of the same time step, or
 * leave the drivable corridor?  obst [n_obst,L,5] = (x, y, length, width, orientation) per obstacle and time step (a static
 * obstacle repeats its row; length <= 0 = absent at that step), or n_obst = 0.  left [n_left,2] / right [n_right,2]: boundary
 * polylines of the corridor in driving direction (0 points = no road check).  Outputs [B]: index of the first offending step,
 * -1 = none.                                                                                                              */
int mpc_validity_batch(mpc_handle* h, int32_t B, int32_t L, const double* traj, double ego_length, double ego_width, int32_t n_obst,
                       const double* obst, int32_t n_left, const double* left, int32_t n_right, const double* right,
                       int32_t* first_collision, int32_t* first_off_road);
int mpc_validity_batch_dev(mpc_handle* h, int32_t B, int32_t L, const double* d_traj, double ego_length, double ego_width, int32_t n_obst,
                           const double* d_obst, int32_t n_left, const double* d_left, int32_t n_right, const double* d_right,
                           int32_t* d_first_collision, int32_t* d_first_off_road, void* stream);
int mpc_closed_loop_batch_dev(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* d_init_state, const double* d_path,
                              const double* d_orient, const double* d_vdes, double* d_traj, double* d_ctrl,
                              int32_t* d_step_status, void* stream);
/* device-pointer forms of the two entry points above: work enqueued on `stream`, no allocation per call (scratch lives in
 * the handle), no synchronisation.  lb / ub / hl / hu of the FORCES solve are small HOST arrays (they go by value).      */
int mpc_metrics_batch_dev(mpc_handle* h, int32_t B, int32_t L, int32_t Lo, const double* d_traj, const double* d_ref_path,
                          const double* d_origin_path, double r_sum, int32_t all_pairs, double* d_deviation, double* d_rmsd,
                          double* d_clearance, void* stream);
int mpc_forces_solve_batch_dev(mpc_handle* h, int32_t B, const double* d_x0, const double* d_xinit, const double* d_all_parameters,
                               const double* lb, const double* ub, const double* hl, const double* hu, int32_t hessian_mode,
                               double* d_x_out, int32_t* d_exitflag, int32_t* d_it, double* d_res, void* stream);

/* Run-time switches of a handle (18).  They are read from the environment once, at mpc_create (MPCGPU_<NAME IN CAPITALS>), and changed afterwards
 * only through this call; value NULL restores the default.  Unknown name -> MPC_ERR_INVALID.
 *   what is solved
 *     "friction_lb"      the lower bound lbg[0] = 0 of the reference's stage-0 friction row sqrt((a_0^2 + v_0^2 tan(delta_0)/2.578)^2)
 *                        (MPC_Planner/optimizer.py:378, 424-425): "nlp" / 0 (default) = implied by the absolute value, no barrier -- a solve returns
 *                        the optimum of the NLP; "ipopt" / 1 = the row as IPOPT sees it, a slack with both bounds and a log barrier on the lower one
 *                        too: the kink of |.| becomes a wall the slack does not cross and a solve can end AT it, depending on the warm start --
 *                        what the reference's recorded ZAM_Over-1_1 run shows at steps 4 and 13 (tests/test_recorded_residuals.py)
 *     "rescue"           0: no second chance for stalled instances (see above)
 *     "rescue_wg"        0: the second chance only behind the launch (rescue_dev); 2: inside k_solve_wg wherever the batch runs one instance per
 *                        workgroup; 1 (default): inside k_solve_wg when the handle's previous solve had stalled instances, else behind the launch
 *                        -- same levels, same bookkeeping, same bits either way (the kernel with the second chance inside costs a batch that
 *                        never stalls 6 - 10 %)
 *     "rescue_alone"     1: the caller knows that instances of this handle's batches stall (collision avoidance): batches up to 8192 instances run in
 *                        k_solve_wg alone, one instance per wavefront, with the second chance inside the launch (B = 4096: 8.3 -> 4.9 ms); an option, not
 *                        a heuristic, because it changes which Riccati sweeps serve an instance -- the last bits of the rows
 *   which kernels serve the iteration loop (every combination gives the same iteration counts; bits as documented in DESIGN.md section 4)
 *     "pipeline"         0: one launch per kernel and iteration (the path of horizons above 63 and trace mode, and what
 *                        an abandoned persistent launch falls back to) instead of the single persistent launch k_pipeline
 *     "hybrid"           0: the pipeline runs every tile to its end; default 1: tiles with few instances left go to k_solve_wg
 *     "hybrid_bx"        instances per wavefront of k_solve_wg: 1, 2, or 0 (default) = by batch size
 *     "hybrid_live"      live instances per tile of 64 at which a tile changes over (a smaller last tile: the same fraction; -1, default: from the
 *                        machine's size; 64: k_solve_wg alone)
 *     "pipe_help"        1 / 0: the Riccati workers of the pipeline take / do not take one published stage item while their tile is with the stage
 *                        workers; -1 (default): where a round has more than three stage items per stage worker (N = 50, B = 8192)
 *     "bound_mask"       0: every bound side looked up at run time (variant 0 of the loop kernels); default 1: the kernels with the bound structure
 *                        of the reference's NLPs compiled in (optimizer.py:421-491) whenever the bounds handed to mpc_set_bounds have it
 *     "big_wg"           1: 512-thread stage workgroups (what horizons above 63 use) also for short horizons
 *     "groups"           2..4: sub-batches on streams of their own (one launch per kernel; measured, not the default)
 *     "max_batch"        instances per chunk (a batch whose workspace would pass 4 GiB is solved in chunks of whole tiles anyway)
 *   host side
 *     "loop_async"       0: closed loop with a host round trip per step
 *     "sync_spin"        0: block in the one synchronisation of a solve instead of polling the stream
 *   states (mpc_get_option only): "pipe_aborts" persistent launches of this handle that had to be abandoned (each such solve started over with one
 *                        launch per kernel), "pipe_disabled" 1 once there were three: the handle stays on one launch per kernel
 *   test and measurement aids
 *     "pipe_test_abort"  1: the persistent launch raises its abort word at once (exercises the restart path)
 *     "pipe_xcd_mask"    pretend XCDs away (a partitioned device)
 *     "poison"           1: NaN into every row of the workspace before a solve (a read of something the solve has not written shows as status -6)
 *     "timing"           sum of 1 (k_stage / k_riccati), 2 (k_pipeline workers), 4 (k_solve_wg rounds), 8 (k_start), 16 (k_solve_wg workgroup trace):
 *                        shader-clock stamps of those kernels printed to stderr after the solve (tools/pipe_timing.py, res_timing.py, ...)       */
int mpc_set_option(mpc_handle* h, const char* name, const char* value);
/* current value of an option (so that a caller that changes one for a moment can put the PREVIOUS value back, not the default) */
int mpc_get_option(const mpc_handle* h, const char* name, int64_t* value);

/* ---- measurement helpers (bench.py / tests) ---------------------------------------------------------- */
/* kernel timing of the LAST mpc_solve_batch[_dev] call, measured with HIP events on the solve stream when
 * profiling is enabled: out[0] = total ms in the Riccati factor/solve kernel, out[1] = its launch count,
 * out[2] = total ms in the stage (line-search/assemble) kernel, out[3] = its launch count,
 * out[4] = ms in init + output kernels, out[5] = IPM iterations launched.                                 */
/* enable = 2: the two kernels of a hybrid solve's iteration loop are timed as ONE span (mpc_get_pipeline_profile out[0] = the loop, the
 * resident profile's out[0] = 0): no event between them -- a marker costs the stream a few microseconds that a production solve does not pay */
int mpc_set_profiling(mpc_handle* h, int32_t enable);
int mpc_get_profile(const mpc_handle* h, double out[6]);
/* When the last call ran all its iterations in ONE persistent launch (k_pipeline: batches of 1024..8192 instances;
 * MPCGPU_PIPELINE=0 turns it off) the per-kernel figures above are zero and this reports instead:
 * out[0] = ms of that launch (profiling enabled only), out[1] = 1 if the pipeline ran (0: one launch per kernel),
 * out[2] = iterations of the slowest tile, out[3] / out[4] = ms the Riccati / stage workers spent waiting for work,
 * summed over workers, out[5] = ms the stage workers spent on work items, out[6] = work items processed,
 * out[7] = stage workers + Riccati workers / 1000.                                                          */
int mpc_get_pipeline_profile(const mpc_handle* h, double out[8]);
/* The workgroup-resident kernels.  k_solve_wg (a workgroup keeps its instances for all their remaining iterations: the stage phases of
 * the other paths + a wave-per-instance Riccati on the fp64 matrix pipe) finishes the instances of the tiles that left the pipeline
 * (hybrid solve, the default: options "hybrid", "hybrid_bx", "hybrid_live") or solves a small batch alone ("hybrid_live" = 64: any batch).
 * out[0] = ms of that launch (profiling enabled only), out[1] = 1 if such a kernel ran in the last call, out[2] = rounds (iterations)
 * of its slowest workgroup, out[3] = workgroups of the launch, out[4] = rounds summed over the workgroups, out[5] = backward Riccati
 * sweeps (> rounds when inertia corrections repeat a sweep; two instances of a wavefront share a sweep), out[6] = instance-iterations
 * it performed (the rest of the call's iterations ran in the pipeline).                                                             */
int mpc_get_resident_profile(const mpc_handle* h, double out[8]);
/* Measurement helper for the roofline of bench.py: GB/s (bytes read + bytes written per second) of a plain streaming copy kernel of
 * this library (16 bytes per lane and access, grid-stride loop over `bytes` of device memory, best of `reps` launches on the handle's
 * stream) -- the bandwidth a kernel that only moves data reaches on this device, next to the 8 TB/s of the data sheet.              */
int mpc_measure_copy_bandwidth(mpc_handle* h, size_t bytes, int32_t reps, double* gbs);

/* debugging: per-iteration per-instance scalars of a host solve.  trace: [max_iter+1, 8, B] doubles
 * rows {mu, theta, phi, alpha, alpha_dual, delta_w, E0, n_trials}; returns iterations launched in *n_it. */
int mpc_solve_batch_trace(mpc_handle* h, int32_t B, const double* x0, const double* p, const double* obst,
                          double* x_out, int32_t* status, int32_t* iters, double* kkt,
                          double* trace, int32_t trace_rows, int32_t* n_it);

int mpc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MPCGPU_H */
