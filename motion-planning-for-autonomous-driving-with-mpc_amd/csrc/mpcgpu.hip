// mpcgpu.hip -- gfx950 (MI355X, CDNA4) kernels and the C-ABI of include/mpcgpu.h.
//
// Replaces, for B independent MPC instances at once, the call
//     res = sol(x0=init_control, p=c_p, lbg=lbg, lbx=lbx, ubg=ubg, ubx=ubx)     MPC_Planner/optimizer.py:607
// i.e. CasADi's nlpsol('ipopt') on the multiple-shooting NLP of optimizer.py:373-558.
//
// Device code (one IPM iteration of an instance = a Riccati sweep of its tile + the stage work of its 8-instance block):
//   stage_block<NX, INIT>  one thread per (instance, horizon stage); a workgroup holds all N+1 stages of `bx`
//                      instances so that per-instance reductions (step lengths, filter line search, KKT error)
//                      stay inside the workgroup: wave-level xor-shuffles across the stages that share a
//                      wavefront, then a small LDS exchange across wavefronts.  Evaluates the bicycle dynamics,
//                      cost, circle-distance rows and their derivatives, condenses slacks/bounds into the stage
//                      Hessian and writes the banded KKT blocks.
//   riccati_tile<NX>   one instance per lane of a compute wave, sequential over the stages: block-tridiagonal
//                      (Riccati) factor + solve of the condensed KKT system, with the sparsity of A_k, B_k hard-wired;
//                      loader waves stream the stage blocks HBM -> LDS by DMA.
//   k_pipeline         ALL iterations of a solve in one persistent launch: tiles cycle between Riccati workers and
//                      stage workers of their XCD (ready queue, arrival counters, bounded waits).
//   k_stage, k_riccati the same two functions as kernels of their own (one launch per kernel and iteration): large
//                      batches, long horizons, trace mode, and the path the pipeline falls back to.
//   k_ingest / k_egest LDS-tiled transposes between the caller's row-major buffers and the tile-major workspace.
// HBM layout: tile-major structure-of-arrays with interleaved row pairs (mpc_prow in mpc_stage_math.h).
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mpc_host_common.h"
#include "mpc_closed_loop.h"
#include "mpc_forces_qp.h"
#include "mpc_riccati_mfma.h"

using namespace mpc;

// ============================================================================================== device
namespace {

// stage-kernel workgroups: 256 threads (one wave per SIMD, the full 512-entry register file, no scratch) when
// bx * (N + 1) <= 256 with bx >= 4 instances (N <= 63), else 512 threads
constexpr int STAGE_MAX_THREADS = 512;

// Workgroup barrier WITHOUT the release fence of __syncthreads(): that fence is `s_waitcnt vmcnt(0)`, i.e. every barrier
// would wait until all of the wave's outstanding global stores are acknowledged by memory (microseconds under load).
// The barriers of these kernels only order LDS traffic, so draining the LDS/scalar counter is all that is needed;
// results written to HBM are consumed by later kernels.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// n values memory -> LDS by a group of NT threads (thread t of it): a thread's loads of one trip are ALL requested before its first LDS write.  (The
// plain loop `for (q = t; q < n; q += NT) lds[q] = mem[q]` compiles to load, s_waitcnt vmcnt(0), ds_write per trip -- one memory round trip per
// trip, eight in a row at the head of k_start.)
template <int MAXJ, class Load, class Store>
__device__ __forceinline__ void batched_fill(const int n, const int t, const int NT, Load ld, Store st) {
    for (int q0 = 0; q0 < n; q0 += MAXJ * NT) {
        double v[MAXJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) { const int q = q0 + j * NT + t; v[j] = (q < n) ? (double)ld(q) : 0.0; }
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) { const int q = q0 + j * NT + t; if (q < n) st(q, v[j]); }
    }
}
// the bounds table [LB | UB] of a batch, nb doubles each
template <int MAXJ>
__device__ __forceinline__ void fill_bounds(const PRef& P, double* tab, const int nb, const int t, const int NT) {
    batched_fill<2 * MAXJ>(2 * nb, t, NT, [&](int q) { return q < nb ? (double)MPC_GP(P.LB, q) : (double)MPC_GP(P.UB, q - nb); }, [&](int q, double v) { tab[q] = v; });
}

// A wave-uniform word read from LDS (a ticket, a work item, a mask another lane published), moved into a SCALAR register.  Not an optimisation:
// left in a vector register, such a value is per-lane state, and when the register allocator parks it (in an AGPR or in scratch) inside a divergent
// region it saves and restores the ACTIVE lanes only -- a lane that was inactive at the save but active when the register was reused in between
// comes back with the register's old content.  Found in round 6 as arrivals credited to the wrong tile by k_pipeline<5, 2, HELP>: the tile number of
// a helper's stage item sat in v209 across stage_block, saved under the mask of the block's running instances, and lane 0 -- whose instance had
// finished -- signalled the arrival with the tile of an EARLIER item (the launch then waits for an arrival that never comes and is abandoned).
// Scalar registers are spilled lane-mask-free.  Everything uniform that lives across stage_block / riccati_tile / wg_stage goes through here.
__device__ __forceinline__ uint32_t lds_uniform(const uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int lds_uniform(const int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double lds_uniform(const double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned long long r = ((unsigned long long)lds_uniform((uint32_t)(u >> 32)) << 32) | lds_uniform((uint32_t)u);
    return __builtin_bit_cast(double, r);
}

// workgroup-wide OR of a predicate with one such barrier (slots double-buffered by call parity)
__device__ __forceinline__ int block_or(int pred, int (*slots)[8], int& parity) {
    const int any = __any(pred) ? 1 : 0;
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) slots[parity][wave] = any;
    lds_barrier();
    int r = 0;
    for (int w = 0; w < nw; ++w) r |= slots[parity][w];
    parity ^= 1;
    return r;
}

template <typename R>
__device__ __forceinline__ R shfl_xor_struct(const R& r, int mask) {
    constexpr int NQ = sizeof(R) / sizeof(double);
    R o;
    const double* s = reinterpret_cast<const double*>(&r);
    double* d = reinterpret_cast<double*>(&o);
#pragma unroll
    for (int i = 0; i < NQ; ++i) d[i] = __shfl_xor(s[i], mask, 64);
    return o;
}

// combine `r` over all stages k of each instance column bl = threadIdx.x % bx; every thread of the column
// ends with the same bits (commutative pairwise combines, identical order in the LDS pass)
template <typename R>
__device__ __forceinline__ void block_reduce(R& r, int bx, double* lds) {
    constexpr int NQ = sizeof(R) / sizeof(double);
    for (int m = bx; m < 64; m <<= 1) {
        const R o = shfl_xor_struct(r, m);
        red_combine(r, o);
    }
    const int nw = blockDim.x >> 6;
    if (nw > 1) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, bl = lane & (bx - 1);
        double* mine = reinterpret_cast<double*>(&r);
        if (lane < bx) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) lds[(wave * NQ + q) * bx + lane] = mine[q];
        }
        lds_barrier();
        R acc;
        double* a = reinterpret_cast<double*>(&acc);
#pragma unroll
        for (int q = 0; q < NQ; ++q) a[q] = lds[q * bx + bl];
        for (int w = 1; w < nw; ++w) {
            R o;
            double* op = reinterpret_cast<double*>(&o);
#pragma unroll
            for (int q = 0; q < NQ; ++q) op[q] = lds[(w * NQ + q) * bx + bl];
            red_combine(acc, o);
        }
        r = acc;
        lds_barrier();
    }
}

// multiplier state that is not touched by the line search is parked in LDS while the search runs (registers of the
// 256-thread variant are capped at 256 so that two workgroups share a CU): rows of STASH_ROWS x blockDim doubles
// state that the line search does not touch is parked in LDS while the search runs (LDS is idle at one workgroup per
// CU; the registers it frees are the difference between a spill-free search loop and scratch traffic)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#ifndef MPC_STAGE_STASH
#define MPC_STAGE_STASH 1
#endif
template <int NX> struct Stash {
    __host__ __device__ static constexpr int rows(bool has_ou) { return 4 * (NX + 2) + 2 * 3 + (has_ou ? 2 * 3 : 0) + 3 + 2 * NX; }
};
template <int NX, bool OUT, uint32_t VM = 0xFFu>
__device__ __forceinline__ void stash_xfer(Ctx<NX>& c, double* st, int T, int t, bool has_ou) {
    int r = 0;
#define MPC_ST(v) do { if (OUT) st[r * T + t] = (v); else (v) = st[r * T + t]; ++r; } while (0)
    // (VM: variables without a bound at any stage keep their compile-time zeros in registers -- nothing to park)
#pragma unroll
    for (int i = 0; i < NX + 2; ++i) { if (!((VM >> i) & 1u)) continue; MPC_ST(c.zl[i]); MPC_ST(c.zu[i]); MPC_ST(c.igl[i]); MPC_ST(c.igu[i]); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { MPC_ST(c.zlo[j]); MPC_ST(c.iglo[j]); }
    if (has_ou) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { MPC_ST(c.zuo[j]); MPC_ST(c.iguo[j]); }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) MPC_ST(c.nuo[j]);
#pragma unroll
    for (int i = 0; i < NX; ++i) { MPC_ST(c.lam[i]); MPC_ST(c.dlam[i]); }
#undef MPC_ST
}

// the variables that carry bounds in the reference's NLP (optimizer.py:421-491: steering rate, acceleration, steering angle, speed):
constexpr uint32_t REF_BOUND_VM = 0x33u;
// variant 2 of k_pipeline / k_solve_wg: that mask, the bounds of its variables at every stage, circle rows with a lower bound only and
// multiplicity 3, one obstacle per batch -- the structure of every NLP the reference builds
constexpr uint32_t REF_DENSE_LO = 0x31u, REF_DENSE_HI = 0x33u;      // (no lower bound on the acceleration)
#ifndef MPC_REF_FLAGS
#define MPC_REF_FLAGS (VM_OSPEC | vm_dense(REF_DENSE_LO, REF_DENSE_HI))
#endif
constexpr uint32_t REF_VM = REF_BOUND_VM | MPC_REF_FLAGS;
// One stage workgroup's share of an iteration: the bx instance columns starting at b0.  `tile_bits` is the activity mask
// of b0's tile (bit l: instance l was iterating when the last Riccati sweep started); a block without such an instance
// leaves before touching HBM.  Called once per launch by k_stage and once per work item by k_pipeline.
template <int NX, bool INIT, int MAXT, uint32_t VM = 0xFFu>
__device__ __forceinline__ void stage_block(const PRef& P, const int n_mult, const int n_z, const int stash_rows, const uint32_t b0,
                                            const unsigned long long tile_bits, double* lds, int (*or_slots)[8], const bool stamp = true,
                                            uint32_t* live_out = nullptr, const bool bounds_in_lds = false) {
    int or_parity = 0;
    Ctx<NX> c;
    constexpr bool STASH = MAXT <= 256 && MPC_STAGE_STASH;
    const bool has_ou = (VM & VM_OSPEC) ? false : (P.has_ou != 0);     // (what the stash parks)
    const int bx = P.bx, t = threadIdx.x;
    c.k = t / bx;
    c.b = (int)b0 + (t & (bx - 1));
    c.valid = (c.k <= P.N) && (c.b < P.B);
    c.active = false;
    c.ill = false;
    c.status = 0;
    c.iters = 0;
    // (hybrid solve: a tile with few instances left is about to go to k_solve_wg -- its items write the mailbox arrays as well, so that the workgroups
    //  there find their instances in the layout they work on; a property of the tile's mask, the same for every item of the round)
    c.mbw = !INIT && P.mbw_live > 0 && __popcll(tile_bits) <= P.mbw_live;
#define MPC_STAMP(i) do { if (P.DBG && t == 0 && stamp) P.DBG[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    MPC_STAMP(0);
    if (!INIT) {
        const unsigned long long m = tile_bits >> (b0 & 63u);
        if ((m & ((bx >= 64) ? ~0ull : ((1ull << bx) - 1ull))) == 0ull) return;
    }
    // LDS: [reduction scratch | bounds table | exchange / stash rows | prefetch images]
    double* lds_b = lds + (blockDim.x >> 6) * 10 * bx;
    const int nb = (P.N + 1) * (NX + 2);
    double* lds_x = lds_b + 2 * nb;
    // (a persistent stage worker of the pipeline copies the table -- the same for every item of the batch -- once: nothing else of
    //  its work items touches that part of the LDS)
    if (!bounds_in_lds)
        fill_bounds<2>(P, lds_b, nb, t, (int)blockDim.x);
    c.bnd = (mpc_lds_cptr)(lds_ptr_t)lds_b;
    c.bnd_ub = nb;
    if (INIT) lds_barrier();
    if (INIT) {
        Red0 r0;
        phase_init_point<NX>(P, c, r0);
        block_reduce(r0, bx, lds);
        phase_init_scalars<NX>(P, c, r0);
    } else {
        phase_load_scalars<NX>(P, c);
        PreTmp<NX> tmp;
        phase_preload<NX, false, VM>(P, c, tmp);           // every array load of the kernel is in flight before the first wait
        phase_premath<NX>(P, c, tmp);
        MPC_STAMP(1);
        // (launches that write the caller's rows themselves: an instance the Riccati sweep has just given up leaves here -- its iterate is in c.z)
        if (P.emit && c.valid && c.status == ST_SWEEP_FAILED) {
            emit_result<NX>(P, c, -7, c.iters, c.k == 0 ? (double)MPC_S(P.SC, SC_E0) : 0.0);
            if (c.k == 0) MPC_S(P.ISC, IS_STATUS) = -7;
        }
        // (the bounds table is read from here on: a block that has just copied it passes a barrier -- behind its loads, which are in flight;
        //  a persistent stage worker that copied it for an earlier item has nothing to wait for)
        if (!bounds_in_lds) lds_barrier();
        // Block-wide "any" WITHOUT a barrier: every wavefront of the block holds a stage thread of EVERY instance column (thread t is stage
        // t / bx of column t % bx, a wavefront covers 64 / bx consecutive stages, and the last wavefront starts at a stage <= N), and all
        // stage threads of an instance hold the same flags (they come out of the block-wide reductions bit for bit) -- so the vote of one
        // wavefront is the vote of the block.  Three barriers + LDS round trips less per work item.
        if (!__any(c.active ? 1 : 0)) return;
        MPC_STAMP(2);
        Red1 r1;
        phase_step_candidates<NX, VM>(P, c, r1);
        MPC_STAMP(3);
        block_reduce(r1, bx, lds);
        phase_linesearch_begin<NX>(P, c, r1);
        MPC_STAMP(4);
        double* stash = lds_x;                                 // shares the exchange region (each thread touches its own column only)
        if (STASH) stash_xfer<NX, true, VM>(c, stash, blockDim.x, t, has_ou);
        while (__any((c.active && c.searching) ? 1 : 0)) {
            Red2 r2;
            phase_trial_eval<NX, VM>(P, c, r2);
            block_reduce(r2, bx, lds);
            phase_linesearch_decide<NX>(P, c, r2);
        }
        if (STASH) stash_xfer<NX, false, VM>(c, stash, blockDim.x, t, has_ou);
        MPC_STAMP(5);
        phase_apply_update<NX, false, VM>(P, c);
        MPC_STAMP(6);
    }
    // neighbour-stage exchange through LDS: thread (k, bl) needs x_{k+1} and lambda_{k+1} of the new iterate
    {
        double* ex = lds_x;                                   // behind the reduction scratch and the bounds table
        const int T = blockDim.x;
#pragma unroll
        for (int i = 0; i < NX; ++i) { ex[i * T + t] = c.z[2 + i]; ex[(NX + i) * T + t] = c.lam[i]; }
        lds_barrier();
        const int tn = t + bx;
        if (tn < T) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { c.xn[i] = ex[i * T + tn]; c.lamn[i] = ex[(NX + i) * T + tn]; }
        }
    }
    MPC_STAMP(7);
    Red3 r3;
    phase_eval_assemble<NX, !INIT, false, VM>(P, c, r3);
    MPC_STAMP(8);
    block_reduce(r3, bx, lds);
    MPC_STAMP(9);
    phase_finish<NX, false>(P, c, r3, n_mult, n_z);
    MPC_STAMP(10);
    // convergence poll without an extra kernel: the stage-0 threads (all in wave 0) count the instances still iterating
    if (live_out != nullptr && t < 64) {
        const unsigned long long m = __ballot((c.valid && c.k == 0 && c.active && c.status == ST_RUNNING) ? 1 : 0);
        if (t == 0) *live_out = (uint32_t)m;
    }
    if (P.run_counter != nullptr && t < 64) {
        const int cnt = __popcll(__ballot((c.valid && c.k == 0 && c.active && c.status == ST_RUNNING) ? 1 : 0));
        if (t == 0 && cnt) atomicAdd(P.run_counter, cnt);
    }
#undef MPC_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
// The stage work of a round of k_solve_wg (ONE wavefront per workgroup: bx = 1 or 2 instances, their stage threads the lanes) -- the same
// phases, bit for bit, as stage_block:
//   mid()          the forward sweep of the KKT solve (called from here so that the stage context is this function's own: declared in the
//                  kernel's scope it cost 270 spilled registers);
//   then           everything the phases read from MEMORY (per-instance scalars, the iterate with its multipliers, the references) is
//                  requested; step and cost-to-go come from the LDS records the sweeps left them in (Rec), then the phases; the new stage blocks go
//                  straight into the records (and to the mailbox copy a repeated sweep rebuilds them from); neighbour stages are
//                  exchanged by lane shuffles; no reduction touches the LDS.
// What the kernel and the phases hand each other besides the records:
//   fail  in : bit g: the sweeps of this round gave instance g up (no admissible inertia correction): its threads see status -7
//   ill   out: bit g: a circle row of instance g carries a large weight (the sticky IS_ILL mark, also written to the workspace)
//   c0    out: LDS, 8 doubles per instance: c_0 = x_0 - r_0 at the new iterate, the start of the next forward sweep
// ---------------------------------------------------------------------------------------------------------------
//   rn, r0   : LDS, what a stage thread reads every round and nothing changes: the reference of its next stage (NX doubles per thread), r_0 (8
//              per instance); fric: may any instance of the workgroup keep its friction row (else its eight scalar rows are not loaded);
//              have: the three are valid (set by the first round after the instances were taken over)
//   scl      : LDS, WgScl::SIZE doubles per instance: the scalars and the filter the phases read every round (filled when the instances are taken over)
struct WgIo { uint32_t fail; uint32_t ill; double* c0; double* rn; double* r0; double* scl; bool fric, have; };
template <int NX, uint32_t VM, class Mid>
__device__ __forceinline__ void wg_stage(const PRef& P, const int ib0, const int ib1, double* lds_bnd, const mpc_lds_ptr rec_base, const int n_mult, const int n_z, const bool stamp,
                                         uint32_t* live_out, WgIo& io, Mid&& mid) {
    const int bx = P.bx, t = threadIdx.x;
    Ctx<NX> c;
    PreTmp<NX> tmp;
    c.k = t / bx;
    c.bl = t & (bx - 1);
    c.mb = (int)blockIdx.x * bx;
    c.b = c.bl ? ib1 : ib0;
    c.valid = (c.k <= P.N) && (c.b >= 0) && (c.b < P.B);
    c.active = false;
    c.ill = false;
    c.status = 0;
    c.iters = 0;
    c.rec = rec_base + ((t & (bx - 1)) * (P.N + 1) + (c.k <= P.N ? c.k : 0)) * Rec<NX>::SIZE;
    c.bnd = (mpc_lds_cptr)(lds_ptr_t)lds_bnd;
    c.bnd_ub = (P.N + 1) * (NX + 2);
    c.scl = (mpc_lds_ptr)(lds_ptr_t)(io.scl + c.bl * WgScl::SIZE);
    // (Measured: with mid() BEHIND the two calls below -- the loads requested under the forward sweep -- the sweep takes 2.9 k ticks longer and
    //  the phases start 2.5 k earlier: what the ~50 loads cost is their ISSUE, ~55 ticks each, wherever it happens, not their latency.)
    mid();
    phase_load_scalars<NX, true>(P, c);
    if (io.have) {
        phase_preload<NX, true, VM, false, true>(P, c, tmp, io.fric);
#pragma unroll
        for (int i = 0; i < NX; ++i) { c.rn[i] = io.rn[t * NX + i]; c.r0[i] = c.k == 0 ? io.r0[c.bl * 8 + i] : 0.0; }
    } else {
        phase_preload<NX, true, VM, false>(P, c, tmp);
#pragma unroll
        for (int i = 0; i < NX; ++i) { io.rn[t * NX + i] = c.rn[i]; if (c.k == 0 && c.valid) io.r0[c.bl * 8 + i] = c.r0[i]; }
        io.fric = __ballot((c.valid && c.k == 0 && c.fric_row) ? 1 : 0) != 0ull;
    }
#define MPC_STAMP(i) do { if (P.DBG && t == 0 && stamp) P.DBG[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    MPC_STAMP(0);
    if (((io.fail >> (t & (bx - 1))) & 1u) && c.valid) { c.status = -7; c.active = false; }      // (what the sweep has written to the status row)
    phase_preload_rec<NX>(P, c, tmp);
    phase_premath<NX>(P, c, tmp);
    MPC_STAMP(1);
    if (!__any(c.active ? 1 : 0)) return;
    MPC_STAMP(2);
    Red1 r1;
    phase_step_candidates<NX, VM>(P, c, r1);
    MPC_STAMP(3);
    block_reduce(r1, bx, nullptr);
    phase_linesearch_begin<NX>(P, c, r1);
    MPC_STAMP(4);
    while (__any((c.active && c.searching) ? 1 : 0)) {
        Red2 r2;
        phase_trial_eval<NX, VM>(P, c, r2);
        block_reduce(r2, bx, nullptr);
        phase_linesearch_decide<NX, true>(P, c, r2);
    }
    MPC_STAMP(5);
    phase_apply_update<NX, true, VM>(P, c);
    MPC_STAMP(6);
    {
        // neighbour-stage exchange: thread (k, bl) needs x_{k+1} and lambda_{k+1} of the new iterate -- lane t + bx of this wavefront
        const int tn = t + bx;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double xs = __shfl_down(c.z[2 + i], (unsigned)bx, 64), ls = __shfl_down(c.lam[i], (unsigned)bx, 64);
            if (tn < 64) { c.xn[i] = xs; c.lamn[i] = ls; }
        }
    }
    MPC_STAMP(7);
    Red3 r3;
    phase_eval_assemble<NX, true, true, VM>(P, c, r3);
    {
        // the instance's stage threads vote; the first of them leaves the (sticky) mark for the MFMA sweeps
        const unsigned long long bal = __ballot((c.active && c.ill) ? 1 : 0);
        const unsigned long long col0 = ~0ull / ((1ull << bx) - 1ull);            // (lanes of instance 0 of the block: every bx-th lane -- bx is a power of two < 64)
        if (c.active && (bal & (col0 << (t & (bx - 1)))) != 0ull && t < bx) MPC_S(P.ISC, IS_ILL) = 1;
        io.ill |= ((bal & col0) != 0ull ? 1u : 0u) | ((bx > 1 && (bal & (col0 << 1)) != 0ull) ? 2u : 0u);
        if (c.k == 0 && c.active) {
#pragma unroll
            for (int i = 0; i < NX; ++i) io.c0[(t & (bx - 1)) * 8 + i] = c.z[2 + i] - c.r0[i];
        }
    }
    MPC_STAMP(8);
    block_reduce(r3, bx, nullptr);
    MPC_STAMP(9);
    phase_finish<NX, true>(P, c, r3, n_mult, n_z);
    MPC_STAMP(10);
    // (the ballot IS the activity mask of the next round -- bit l: instance b0 + l goes on; no status row is re-read)
    {
        const unsigned long long m = __ballot((c.valid && c.k == 0 && c.active && c.status == ST_RUNNING) ? 1 : 0);
        if (t == 0) *live_out = (uint32_t)m;
    }
#undef MPC_STAMP
}

// VM: bound structure compiled into the phases (0xFF: looked up at run time; REF_VM: the reference's -- every path of a handle uses the
// same instantiation of the phases, so that the pipeline, its fallback of one launch per kernel and the closed loop's replay give the same bits)
template <int NX, bool INIT, int MAXT, uint32_t VM = 0xFFu>
__global__ void __launch_bounds__(MAXT, (INIT && MAXT <= 256) ? 2 : 1) k_stage(const Params Pk, const int n_mult, const int n_z, const int stash_rows) {
    const PRef P(Pk);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int or_slots[2][8];
    // workgroups are dealt round-robin to the 8 XCDs: renumber so that each XCD gets a contiguous run of instance
    // columns -- the 64/bx workgroups that share every 128-byte line of a tile then share one L2
    uint32_t blk = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
    const uint32_t b0 = (blk + (uint32_t)P.tile0 * (64u / (uint32_t)P.bx)) * (uint32_t)P.bx;
    // scalar load, uniform branch: finished workgroups leave without any vector memory traffic
    const unsigned long long bits = (!INIT && P.tile_mask != nullptr) ? P.tile_mask[b0 >> 6] : ~0ull;
    stage_block<NX, INIT, MAXT, VM>(P, n_mult, n_z, stash_rows, b0, bits, lds, or_slots);
}

// ---------------------------------------------------------------------------------------------------------------
// Riccati sweep (riccati_tile; k_riccati is the same as a kernel of its own): block-tridiagonal factor + solve.  One tile of
// 64 instances = three wavefronts (the second loader, wave 2, only works in the forward sweep):
//   wave 1 (loader)  streams the condensed stage blocks HBM -> LDS with asynchronous buffer->LDS DMA (1 KiB per wave
//                    instruction, no VGPR round trip) RIC_DEPTH-1 stages ahead into a ring of RIC_DEPTH slots; its
//                    vmcnt counter tracks nothing but those DMAs, so "stage k has landed" is an exact s_waitcnt.
//   wave 0 (compute) one instance per lane: s_barrier -> ds_read of the stage block -> ~270 fp64 operations -> 16-byte
//                    fire-and-forget stores of gains / cost-to-go; it never waits on HBM.
// The stage block of stage k is ONE contiguous chunk of the tile-major workspace (whole row pairs, 1 KiB each), so the LDS
// image is [row pair][lane][2] and a lane reads both rows of a pair with one conflict-free ds_read_b128.  Arithmetic: riccati_backward_step / riccati_forward_step of
// mpc_stage_math.h (shared with the CPU emulation harness).  Inertia correction: if some lane finds an indefinite
// 2x2 block the wave repeats the sweep with delta_w added for those lanes (flag through LDS keeps the loader in step).
// ---------------------------------------------------------------------------------------------------------------
constexpr int32_t HO_IN_MB = 0x40000000;  // flag in an entry of the hand-over lists
constexpr uint32_t HO_BUCKETS = 4;  // hand-over lists of the hybrid solve, by KKT error: >= 1e-3 | >= 1e-4 | >= 1e-5 | below (rank correlation with the iterations left: 0.9)
constexpr int RIC_DEPTH = 4;        // backward ring (stage blocks, 17 KiB each at nx = 6)
constexpr int RIC_DEPTH_F = 10;     // forward ring (gains + A + defect rows, 13 KiB each): stages are short, so look further ahead --
                                    // fed by TWO loader waves (even / odd stages), each limited to 4 stages in flight by the 6-bit vmcnt

template <int NCH>
__device__ __forceinline__ void wait_dma_behind(int stages_behind) {
    // wait until at most `stages_behind` stage-DMAs (NCH buffer ops each) are still in flight
    if (stages_behind >= 4 && 4 * NCH <= 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NCH <= 63 ? 4 * NCH : 0) : "memory");
    else if (stages_behind >= 3 && 3 * NCH <= 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NCH <= 63 ? 3 * NCH : 0) : "memory");
    else if (stages_behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NCH) : "memory");
    else if (stages_behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The tiles of the hybrid solve leave the pipeline with a few instances still iterating: they go onto the lists k_solve_wg deals its
// workgroups from -- one list per bucket of the KKT error the instance stands at, a predictor of the iterations it still needs (one
// agent-scope ticket per tile and bucket, a slot per live lane).  Done by the pipeline's stage workers on their way out (option ho_inline,
// the default) or by a launch of its own between the two kernels (5.5 us).  (Appended by the retiring Riccati worker inside riccati_tile
// the few instructions cost its sweeps 13 us per solve; as an out-of-line call they put the whole kernel on scratch.)
// (one wavefront, lane = instance b of a tile)
__device__ __forceinline__ void ho_lists(const PRef& P, const uint32_t bb, int32_t* ho_list, uint32_t* ho_count, const uint32_t* tile_word) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x & 63;
    const bool active = (int)bb < P.B && (int32_t)MPC_U(P.ISC, (uint32_t)IS_STATUS) == ST_RUNNING;
    if (__ballot(active ? 1 : 0) == 0ull) return;
    // (bit 30 of an entry: the instance's rows are in the mailbox arrays already -- bit 31 of its tile's arrival counter, see k_pipeline)
    const int32_t in_mb = (tile_word != nullptr && (__hip_atomic_load(tile_word + (bb >> 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 31)) ? HO_IN_MB : 0;
    const double e0 = active ? (double)MPC_U(P.SC, (uint32_t)SC_E0) : 0.0;
    const int bucket = e0 >= 1e-3 ? 0 : e0 >= 1e-4 ? 1 : e0 >= 1e-5 ? 2 : 3;
    for (int q = 0; q < (int)HO_BUCKETS; ++q) {
        const unsigned long long mq = __ballot((active && bucket == q) ? 1 : 0);
        if (mq == 0ull) continue;
        uint32_t base = 0u;
        if (lane == 0) base = __hip_atomic_fetch_add(ho_count + q, (uint32_t)__popcll(mq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (active && bucket == q) ho_list[(uint32_t)q * (uint32_t)P.Bp + base + (uint32_t)__popcll(mq & ((1ull << lane) - 1ull))] = (int32_t)bb | in_mb;
    }
#endif
}

// the Riccati factor + solve of one tile of 64 instances by three wavefronts; returns the tile's activity mask (0: no
// instance of the tile is iterating, nothing was touched)
template <int NX>
__device__ __forceinline__ unsigned long long riccati_tile(const PRef& P, const uint32_t tile, char* smem, const bool stamp = true, const int handover_live = 0) {
#if defined(__HIP_DEVICE_COMPILE__)      // device-only builtins (buffer->LDS DMA, readfirstlane)
    using D = Dim<NX>;
    constexpr int NS = D::NS;
    constexpr uint32_t BLK_BYTES = MPC_EV(D::NBLK) * 512u;              // a stage block: whole row pairs, 1 KiB each ([pair][lane][2])
    constexpr int BLK_CHUNKS = (BLK_BYTES + 1023u) / 1024u;
    constexpr uint32_t SLOT = BLK_CHUNKS * 1024u;
    constexpr int KK_CHUNKS = (D::NKK * 512u + 1023u) / 1024u;
    constexpr int FWD_CHUNKS = KK_CHUNKS + 6;                      // gains + A rows + defect rows
    constexpr uint32_t FSLOT = FWD_CHUNKS * 1024u;
    static_assert(2 * BLK_CHUNKS <= 63 && RIC_DEPTH == 4 && 4 * FWD_CHUNKS <= 63 && RIC_DEPTH_F == 10, "vmcnt is a 6-bit counter");
    constexpr uint32_t RING_BYTES = (RIC_DEPTH * SLOT > RIC_DEPTH_F * FSLOT) ? RIC_DEPTH * SLOT : RIC_DEPTH_F * FSLOT;
    int* flag = reinterpret_cast<int*>(smem + RING_BYTES);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int b = (int)(tile * 64u) + lane;
    const uint32_t bb = (uint32_t)b;
    const int N = P.N;
    const bool active = (b < P.B) && ((int32_t)MPC_U(P.ISC, (uint32_t)IS_STATUS) == ST_RUNNING);
    const unsigned long long act_mask = __ballot(active ? 1 : 0);
    if (act_mask == 0ull) return 0ull;                              // all waves see the same 64 instances
    // hybrid solve: a tile with this few instances left is no longer worth a 31-stage pass of a whole wavefront -- it leaves the
    // pipeline untouched, k_solve_wg (one wavefront per instance, MFMA Riccati) finishes its instances behind this launch
    // (a threshold per tile -- the same FRACTION of its population: the last tile of a batch that is not a multiple of 64 would otherwise leave at
    //  once when it holds no more than the count, and its instances would do all their iterations behind the pipeline)
    //  (rounded up: a tile of one to four instances still leaves at once -- it would otherwise keep the whole launch going for its last instance)
    if (__popcll(act_mask) * 64 < handover_live * min(64, P.B - (int)(tile * 64u)) + 64) return 0ull;
#define RIC_STAMP(i) do { if (P.DBG && threadIdx.x == 0 && stamp) P.DBG[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    RIC_STAMP(0);
    const __amdgpu_buffer_rsrc_t rsrc = P.rws;
    const uint32_t tile_off = tile * P.tile_elems * 8u;
    const uint32_t blk_base = (uint32_t)(uintptr_t)P.BLK - (uint32_t)(uintptr_t)P.WS + tile_off;
    const uint32_t kk_base = (uint32_t)(uintptr_t)P.KK - (uint32_t)(uintptr_t)P.WS + tile_off;
    auto dma = [&](uint32_t src_, uint32_t dst_, int nchunks) {
        const uint32_t src = (uint32_t)__builtin_amdgcn_readfirstlane((int)src_), dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)dst_);
#pragma unroll
        for (int c = 0; c < nchunks; ++c)
        {   // (P.xcu: the rows were written by stage workers on other CUs of this XCD in this launch -- sc1, past this CU's vector L1; see DevParams)
            if (P.xcu) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + dst + (uint32_t)c * 1024u), 16, lane * 16, (int)(src + (uint32_t)c * 1024u), 0, MPC_AUX_SC1);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + dst + (uint32_t)c * 1024u), 16, lane * 16, (int)(src + (uint32_t)c * 1024u), 0, 0);
        }
    };
    // row r of an LDS image that starts at `off` (the DMA copies the row pairs verbatim)
    auto lds_r = [&](uint32_t off, uint32_t r) { return *reinterpret_cast<const double*>(smem + off + mpc_prow(r) * 8u + (uint32_t)lane * 16u); };
    auto read_stage = [&](uint32_t slot, RicStage<NX>& s) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = i; j < NX; ++j)
                s.H[D::sidx(i, j)] = (D::hrow(i, j) >= 0) ? lds_r(slot, D::B_H + (D::hrow(i, j) >= 0 ? D::hrow(i, j) : 0)) : 0.0;
        }
        s.ruu[0] = lds_r(slot, D::B_RUU);
        s.ruu[1] = lds_r(slot, D::B_RUU + 1);
#pragma unroll
        for (int i = 0; i < 6; ++i) s.a[i] = lds_r(slot, D::B_A + i);
#pragma unroll
        for (int i = 0; i < NX; ++i) { s.gx[i] = lds_r(slot, D::B_GX + i); s.cn[i] = lds_r(slot, D::B_CN + i); }
        s.gu[0] = lds_r(slot, D::B_GU);
        s.gu[1] = lds_r(slot, D::B_GU + 1);
    };
    // ================================================================ backward sweep(s)
    double delta = 0.0, delta_last = 0.0, hux0 = 0.0, hux1 = 0.0;
    bool need = false, failed = false;
    if (wave == 0) {
        need = active;
        if (active) { delta_last = MPC_U(P.SC, (uint32_t)SC_DLAST); hux0 = MPC_U(P.SC, (uint32_t)SC_HUX0); hux1 = MPC_U(P.SC, (uint32_t)SC_HUX1); }
    }
    for (;;) {
        if (wave >= 2) {
            for (int t = 0; t <= N; ++t) lds_barrier();             // second loader: idle in the backward sweep (a fourth wave -- k_pipeline's helping workers -- in both)
        } else if (wave == 1) {
            // ---------------- loader: stages N, N-1, ... ; stage N-t lives in slot t % RIC_DEPTH
            for (int j = 0; j < RIC_DEPTH - 1 && j <= N; ++j) dma(blk_base + (uint32_t)(N - j) * BLK_BYTES, (uint32_t)j * SLOT, BLK_CHUNKS);
            for (int t = 0; t <= N; ++t) {
                const int k = N - t;
                wait_dma_behind<BLK_CHUNKS>(k < RIC_DEPTH - 2 ? k : RIC_DEPTH - 2);       // stage k has landed
                lds_barrier();                                                           // compute: go on stage k (and is done with k+1)
                const int kn = k - (RIC_DEPTH - 1);
                if (kn >= 0) dma(blk_base + (uint32_t)kn * BLK_BYTES, (uint32_t)((t + RIC_DEPTH - 1) % RIC_DEPTH) * SLOT, BLK_CHUNKS);
            }
        } else {
            // ---------------- compute
            bool ok = need;
            // symmetrised G'K for the instances that ever needed an inertia correction (mpc_stage_math.h, ric_matrix_step); a
            // wavefront without any runs the sweep instantiated without the term (decided per sweep, so that the loop of the
            // common case is the loop it always was)
            const bool sym = delta != 0.0 || delta_last != 0.0;
            auto sweep = [&](auto sym_tag, auto ne_tag) {
                constexpr bool SYM = decltype(sym_tag)::value;
                constexpr int NE = decltype(ne_tag)::value;
                double Ps[NS], pv[NX];
                for (int t = 0; t <= N; ++t) {
                    const int k = N - t;
                    if (t == 15) RIC_STAMP(3);
                    lds_barrier();
                    if (t == 15) RIC_STAMP(4);
                    RicStage<NX> s;
                    read_stage((uint32_t)(t % RIC_DEPTH) * SLOT, s);
                    if (t == 0) {
#pragma unroll
                        for (int i = 0; i < NS; ++i) Ps[i] = s.H[i];
#pragma unroll
                        for (int i = 0; i < NX; ++i) { Ps[D::sidx(i, i)] += delta; pv[i] = s.gx[i]; }
                        if (ok) {
                            double pk[D::NPK];
#pragma unroll
                            for (int i = 0; i < NS; ++i) pk[i] = Ps[i];
#pragma unroll
                            for (int i = 0; i < NX; ++i) pk[NS + i] = pv[i];
                            ws_store_rows<D::NPK>(MPC_ROWS(MPC_UK(P.PK, D::NPK, N, e)), pk);
                        }
                    } else if (ok) {
                        ok = riccati_backward_step<NX, NE, SYM>(P, bb, k, s, delta, hux0, hux1, Ps, pv, SYM && sym);
                    }
                    if (t == 15) RIC_STAMP(5);
                }
            };
            // (six states with a costless, unbounded progress state -- flagged by the host -- and no inertia correction anywhere in the wavefront:
            //  the recursion runs on five states, row and column of the sixth stay the zeros they are; same bits, a quarter fewer instructions)
            if (__any(sym ? 1 : 0)) sweep(std::true_type{}, std::integral_constant<int, NX>{});
            else if (NX == 6 && P.dec_s) sweep(std::false_type{}, std::integral_constant<int, (NX == 6 ? 5 : NX)>{});
            else sweep(std::false_type{}, std::integral_constant<int, NX>{});
            if (need && ok) need = false;
            else if (need) {
                if (delta == 0.0) delta = (delta_last == 0.0) ? DW_0 : fmax(DW_MIN, KW_MINUS * delta_last);
                else delta *= (delta_last == 0.0) ? KW_PLUS_BAR : KW_PLUS;
                if (delta > DW_MAX) { need = false; failed = true; }
            }
            const int again = __any(need ? 1 : 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               // gains / cost-to-go are out
            if (lane == 0) *flag = again;
        }
        lds_barrier();
        const int again = lds_uniform(*flag);
        lds_barrier();
        if (!again) break;
    }
    RIC_STAMP(1);
    // ================================================================ forward sweep
    const bool go = (wave == 0) && active && !failed;
    if (wave >= 3) {
        for (int k = 0; k < N; ++k) lds_barrier();
    } else if (wave >= 1) {
        // two loaders: wave 1 owns the even stages, wave 2 the odd ones.  Stage k lives in slot k % RIC_DEPTH_F; the slot
        // of stage k-1 is free once the compute wave has reached barrier k, and gets stage k-1+RIC_DEPTH_F.
        const int par = wave - 1;
        auto dma_fwd = [&](int k, uint32_t dst) {
            dma(kk_base + (uint32_t)k * (D::NKK * 512u), dst, KK_CHUNKS);
            static_assert(D::B_A % 2 == 0 && D::B_CN % 2 == 0 && D::NKK % 2 == 0, "sub-blocks copied on their own start on a row-pair boundary");
            dma(blk_base + (uint32_t)k * BLK_BYTES + D::B_A * 512u, dst + KK_CHUNKS * 1024u, 3);
            dma(blk_base + (uint32_t)k * BLK_BYTES + D::B_CN * 512u, dst + (KK_CHUNKS + 3) * 1024u, 3);
        };
        for (int j = par; j < RIC_DEPTH_F - 1 && j < N; j += 2) dma_fwd(j, (uint32_t)j * FSLOT);
        for (int k = 0; k < N; ++k) {
            if ((k & 1) == par) {
                const int last = (N - 1 < k + RIC_DEPTH_F - 2) ? N - 1 : k + RIC_DEPTH_F - 2;    // newest stage requested so far
                wait_dma_behind<FWD_CHUNKS>((last - k) >> 1);                                     // my stages after k in flight
            }
            lds_barrier();
            const int kn = k + RIC_DEPTH_F - 1;
            if (kn < N && (kn & 1) == par) dma_fwd(kn, (uint32_t)(kn % RIC_DEPTH_F) * FSLOT);
        }
    } else {
        if (active && failed) MPC_U(P.ISC, (uint32_t)IS_STATUS) = P.emit ? ST_SWEEP_FAILED : -7;
        if (go) {
            if (delta > 0.0) MPC_U(P.SC, (uint32_t)SC_DLAST) = delta;
            MPC_U(P.SC, (uint32_t)SC_DELTA) = delta;
        }
        double dx[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[i] = go ? -(double)MPC_U(P.SC, (uint32_t)(SC_C0 + i)) : 0.0;
        for (int k = 0; k < N; ++k) {
            const uint32_t slot = (uint32_t)(k % RIC_DEPTH_F) * FSLOT;
            if (k == 15) RIC_STAMP(6);
            lds_barrier();
            if (k == 15) RIC_STAMP(7);
            if (k == 0) RIC_STAMP(9);
            FwdStage<NX> f;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                f.K0[j] = lds_r(slot, (uint32_t)j);
                f.K1[j] = lds_r(slot, (uint32_t)(NX + j));
                f.cn[j] = lds_r(slot + (KK_CHUNKS + 3) * 1024u, (uint32_t)j);
            }
            f.kf0 = lds_r(slot, (uint32_t)(2 * NX));
            f.kf1 = lds_r(slot, (uint32_t)(2 * NX + 1));
#pragma unroll
            for (int i = 0; i < 6; ++i) f.a[i] = lds_r(slot + KK_CHUNKS * 1024u, (uint32_t)i);
            // executed by every lane (finished / padding instances just write an unused step): keeping the stores out of
            // a divergent branch spares a waterfall loop around each of them
            riccati_forward_step<NX>(P, bb, k, f, dx);
            if (k == 15) RIC_STAMP(8);
        }
        if (go) {
            double dz[D::NZ];
            dz[0] = 0.0;
            dz[1] = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) dz[2 + i] = dx[i];
            ws_store_rows<D::NZ>(MPC_ROWS(MPC_UK(P.DZ, D::NZ, N, e)), dz);
        }
    }
    RIC_STAMP(2);
#undef RIC_STAMP
    return act_mask;
#else
    return 0ull;
#endif
}

template <int NX>
__global__ void __launch_bounds__(192) k_riccati(const Params Pk) {
    const PRef P(Pk);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t tile = blockIdx.x + (uint32_t)P.tile0;
    const unsigned long long act_mask = riccati_tile<NX>(P, tile, smem);
    // (every lane holds the same mask; the stage kernel that follows reads it with a scalar load)
    if (P.tile_mask != nullptr && threadIdx.x == 0) P.tile_mask[tile] = act_mask;
}

// ---------------------------------------------------------------------------------------------------------------
// k_pipeline: ALL iterations of a solve in one launch.  k_riccati is the latency of one 31-stage chain on 64 of the 256
// CUs, k_stage needs two rounds of whole-CU workgroups; launched back to back, each waits for the other.  Here the two
// run side by side as roles of one persistent grid (one workgroup per CU) and every tile of 64 instances cycles on its
// own:   Riccati worker(tile) --ready queue--> 64/bx stage work items --per-tile arrival counter--> Riccati worker ...
// so the stage workers always have the items of SOME tile to chew on while the others sit in their Riccati chain, and a
// tile whose instances have all converged simply stops producing items (no batch-wide convergence poll).
//
// Hand-off protocol.  A tile's data never leaves one XCD: a workgroup reads the XCD it actually runs on
// (HW_REG_XCC_ID, not a guess from blockIdx) and serves only the tiles with tile % 8 == that XCD, in the role its
// arrival order on that XCD gives it.  Producer and consumer of every row therefore share one L2:
//   producer: plain stores -> s_waitcnt vmcnt(0) (acknowledged by the L2) -> barrier -> agent-scope atomic on the flag
//   consumer: relaxed agent-scope poll by one lane -> barrier -> loads, EVERY ONE of them with the sc1 bit (DevParams::xcu: served by the L2,
//             never by this CU's vector L1 -- a CU's L1 is not refreshed by another CU's stores, and neither `buffer_inv sc0` nor a workgroup-
//             scope fence drops its lines: measured, profiles/r05_store_pairing.txt, tools/ubench/stale_l1.hip).  No invalidate, no assumption
//             about what the L1 holds; tests/test_isa_hazards.py proves from the ISA that no buffer load of the kernel lacks the bit.
// The flags themselves (queue slots, counters, abort word) are agent-scope atomics, valid across XCDs.
// Every spin is bounded: a wait longer than PIPE_SPIN_LIMIT sets the abort word, every worker leaves, and the host
// re-runs the solve with one launch per kernel (also if the dispatcher leaves an XCD without stage workers).
// ---------------------------------------------------------------------------------------------------------------
struct PipeArgs {
    uint32_t* ctl;          // control block, zeroed before every launch (layout below)
    uint32_t ntiles;
    uint32_t xcd_mask;      // bit x: the device has an XCD with HW_REG_XCC_ID x (tile t belongs to the (t mod n)-th of them)
    uint32_t n_ric;         // Riccati workers per XCD (the first arrivals)
    uint32_t cap;           // ready-queue slots per XCD, a power of two >= 2 * items of one XCD
    uint32_t items;         // stage work items per tile = 64 / bx
    uint32_t handover;      // a tile with at most this many instances still iterating leaves the pipeline (0: tiles run to the end)
    int32_t* ho_list;       // ... and the stage workers put them onto the hand-over lists on their way out (nullptr: k_ho_lists does, or nobody)
    uint32_t flags;         // bit 1: raise the abort word at once (option pipe_test_abort: exercises the host's restart path)
};
constexpr uint32_t PIPE_X_STRIDE = 64;          // uint32 words per XCD record: arrive @0, head @16, tail @32, finished @48, hand-over ticket @56
constexpr uint32_t PIPE_ABORT = 8 * PIPE_X_STRIDE;      // abort word; +1 rounds (max), +2.. statistics
constexpr uint32_t PIPE_STATS = PIPE_ABORT + 2;         // [wait ticks riccati, wait ticks stage, busy ticks stage, items, workers stage, workers riccati]
constexpr uint32_t PIPE_WG = PIPE_ABORT + 16;           // statistics of the k_solve_wg launch behind the pipeline (5 words: rounds max, rounds, sweeps, instance-rounds, rescued): zeroed and copied back with the block
constexpr uint32_t PIPE_HO = PIPE_ABORT + 24;           // hand-over list: instances left by retiring tiles, per bucket of their KKT error (HO_BUCKETS words)
constexpr uint32_t PIPE_HDR = PIPE_ABORT + 32;          // then: stage_done[ntiles] | pad to 2 words | slots[8][cap] (uint64)
constexpr uint32_t PIPE_EXIT = 0xFFFFFFFFu;
constexpr unsigned long long PIPE_SPIN_LIMIT = 5000000ull;      // 100 MHz wall-clock ticks = 50 ms
#ifndef MPC_PIPE_SLEEP
#define MPC_PIPE_SLEEP 8                                        // s_sleep argument between two polls of a hand-off flag (x 64 clocks)
#endif
__host__ __device__ inline uint32_t pipe_slots_off(uint32_t ntiles) { return (PIPE_HDR + ntiles + 1u) & ~1u; }
__host__ __device__ inline size_t pipe_ctl_words(uint32_t ntiles, uint32_t cap) { return (size_t)pipe_slots_off(ntiles) + (size_t)16 * cap; }

// acquire side of a hand-off: nothing for the hardware to do -- every workspace load of the kernel is an sc1 load (DevParams::xcu), served by the
// XCD's L2, where the producer's rows are once its `s_waitcnt vmcnt(0)` has returned; this keeps the compiler from moving loads above the flag
__device__ __forceinline__ void pipe_acquire() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}
__device__ __forceinline__ uint32_t pipe_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pipe_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t pipe_add(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// measurement helper (mpc_measure_copy_bandwidth): the plain streaming copy, 16 bytes per lane and access, four independent accesses
// per lane and trip (a workgroup moves 16 KiB per trip), grid-stride
__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 1024;
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + 768 < n; i += stride) {
        const uint4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
    }
    for (; i < n; i += 256) dst[i] = src[i];
}

// debugging aid (option poison): rows [r0, r1) of every tile of the double workspace set to NaN before a solve -- a read of memory the solve has not
// written shows as status -6 instead of as a stale but plausible number
__global__ void k_poison(double* ws, uint32_t tile_elems, uint32_t ntiles, uint32_t e0, uint32_t e1) {
    const size_t n = (size_t)(e1 - e0) * ntiles;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t tile = i / (e1 - e0), e = e0 + i % (e1 - e0);
        ws[tile * tile_elems + e] = __builtin_nan("");
    }
}

// which XCDs does this device have?  (one bit per HW_REG_XCC_ID that some workgroup of a grid of 4 x CUs ran on)
__global__ void k_xcd_census(uint32_t* mask) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x == 0) atomicOr(mask, 1u << ((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u));
#endif
}

// VAR: 0 bounds looked up at run time; 2 the bound structure of the reference compiled in (REF_BOUND_VM: only steering rate, acceleration,
// steering angle and speed carry bounds -- the sides of the other variables, their multipliers and 1/gap registers vanish from the code)
// HELP: the Riccati workers take a stage item while their tile is with the stage workers (below) -- a variant of its own: the extra code in the
// Riccati worker's loop costs the sweeps of the default kernel 3.5 % through its register allocation, and only batches whose stage items outnumber
// the stage workers three to one gain from it (N = 50: 16 items per tile)
template <int NX, int VAR, bool HELP = false>
__global__ void __launch_bounds__(256) k_pipeline(const Params Pk, const PipeArgs A, const int n_mult, const int n_z, const int stash_rows) {
    const PRef P = mpc_pref(Pk, true);  // (xcu: rows travel between the CUs of an XCD inside this launch -- every workspace load is an sc1 load)
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int or_slots[2][8];
    __shared__ uint32_t sh_word[4];
    const int t = threadIdx.x;
    // the XCD this workgroup runs on, as an index among the XCDs the device was seen to have (k_xcd_census at handle
    // creation: 8 on a whole MI355X, fewer in a partitioned mode); a workgroup on an XCD outside that set has no tiles
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;       // HW_REG_XCC_ID[3:0]
    if (!((A.xcd_mask >> xcc) & 1u)) return;
    const uint32_t n_xcd = (uint32_t)__popc(A.xcd_mask), xcd = (uint32_t)__popc(A.xcd_mask & ((1u << xcc) - 1u));
    uint32_t* X = A.ctl + xcd * PIPE_X_STRIDE;
    uint32_t* abort_w = A.ctl + PIPE_ABORT;
    uint32_t* stage_done = A.ctl + PIPE_HDR;
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(A.ctl + pipe_slots_off(A.ntiles)) + (size_t)xcd * A.cap;
    const uint32_t n_tiles_x = A.ntiles > xcd ? (A.ntiles - xcd + n_xcd - 1u) / n_xcd : 0u;   // tiles xcd, xcd + n_xcd, ...
    if (n_tiles_x == 0u) return;
    if ((A.flags & 2u) && t == 0) pipe_st(abort_w, 1u);
    if (t == 0) sh_word[0] = pipe_add(X + 0, 1u);
    lds_barrier();
    const uint32_t slot = lds_uniform(sh_word[0]);
    const uint32_t n_ric = A.n_ric < n_tiles_x ? A.n_ric : n_tiles_x;
    lds_barrier();
    unsigned long long waited = 0;
    // (profiling aid: stamps of the worker's 6th pass / item only -- later ones would overwrite them)
    uint32_t n_pass = 0;
#define PIPE_STAMP(i) do { if (P.DBG && t == 0 && n_pass == 5u) P.DBG[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    if (slot < n_ric) {
        // ============================================================ Riccati worker: local tiles slot, slot + n_ric, ...
        // (option pipe_help: a Riccati worker whose tile has just gone to the stage workers takes ONE stage item of the queue itself instead of
        //  idling through the item time -- a third more stage capacity per XCD; its fourth wavefront then stays, idle, through the sweeps)
        constexpr bool helper = HELP;
        if (!helper && t >= 192) return;
        uint32_t n_help = 0u;
        const uint32_t n_own = (n_tiles_x - slot + n_ric - 1u) / n_ric;
        uint32_t fin = 0u, round = 0u;
        for (;; ++round) {
            bool all_done = true;
            for (uint32_t j = 0; j < n_own; ++j) {
                if ((fin >> j) & 1u) continue;
                all_done = false;
                const uint32_t tile = (slot + j * n_ric) * n_xcd + xcd;
                PIPE_STAMP(11);
                if (t == 0) {
                    uint32_t ok = 1u;
                    const uint32_t need = A.items * round;
                    const unsigned long long t0 = wall_clock64();
                    while (pipe_ld(stage_done + tile) < need) {
                        if (pipe_ld(abort_w)) { ok = 0u; break; }
                        if (wall_clock64() - t0 > PIPE_SPIN_LIMIT) {
                            // (diagnostics of the abort message: which tile and round, the arrivals seen, the arrivals a millisecond later, the ticks waited)
                            pipe_st(abort_w + 28, (tile << 16) | (round & 0xFFFFu)); pipe_st(abort_w + 29, pipe_ld(stage_done + tile)); pipe_st(abort_w + 31, (uint32_t)(wall_clock64() - t0));
                            const unsigned long long t1 = wall_clock64();
                            while (wall_clock64() - t1 < 100000ull) __builtin_amdgcn_s_sleep(8);
                            pipe_st(abort_w + 30, pipe_ld(stage_done + tile));
                            pipe_st(abort_w, 2u); ok = 0u; break;
                        }
                        __builtin_amdgcn_s_sleep(MPC_PIPE_SLEEP);
                    }
                    waited += wall_clock64() - t0;
                    pipe_acquire();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the invalidate has completed before the barrier lets the other waves load
                    sh_word[1] = ok;
                }
                lds_barrier();
                if (lds_uniform(sh_word[1]) == 0u) return;
                PIPE_STAMP(12);
                const unsigned long long mask = riccati_tile<NX>(P, tile, reinterpret_cast<char*>(lds), n_pass == 5u, (int)A.handover);
                if (mask == 0ull) {
                    fin |= 1u << j;
                    if (t == 0) {
                        // (bit 31 of the tile's arrival counter, for k_solve_wg: the items of its last round wrote the mailbox arrays -- same predicate as stage_block's)
                        if (round > 0u && P.mbw_live > 0 && __popcll(__hip_atomic_load(P.tile_mask + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <= P.mbw_live)
                        {   __hip_atomic_fetch_or(stage_done + tile, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (performed before this worker counts its tile as finished: the stage workers read it on their way out)
                        }
                        atomicMax(A.ctl + PIPE_ABORT + 1, round);
                        pipe_add(X + 48, 1u);
                    }
                } else if (t < 64) {
                    // (the tile's mask goes out with the sweep's last stores: ONE wait covers both -- it is read by a stage worker only after it has seen
                    //  its queue slot, which is written behind that wait)
                    if (t == 0) __hip_atomic_store(P.tile_mask + tile, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // gains, cost-to-go and the step are in the L2
                    if (t == 0) {
                        const uint32_t tk = pipe_add(X + 32, A.items);
                        for (uint32_t q = 0; q < A.items; ++q)
                            __hip_atomic_store(slots + ((tk + q) & (A.cap - 1u)), ((unsigned long long)(tk + q + 1u) << 32) | (unsigned long long)((tile << 8) | q),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                PIPE_STAMP(13);
                ++n_pass;
                lds_barrier();                     // the rings are free again
                if constexpr (helper) if (mask != 0ull) {
                    if (t == 0) {
                        uint32_t item = PIPE_EXIT;
                        unsigned long long bits = 0ull;
                        // the tile this worker turns to next: still with the stage workers?  (else its Riccati pass goes first)
                        uint32_t jn = j, rn = round;
                        do { if (++jn == n_own) { jn = 0u; ++rn; } } while (((fin >> jn) & 1u) && jn != j);
                        const uint32_t tile_n = (slot + jn * n_ric) * n_xcd + xcd;
                        if (pipe_ld(stage_done + tile_n) < A.items * rn) {
                            // a ticket only for an item that has been published (the stage workers draw theirs blindly and wait; this worker must not)
                            uint32_t hd = pipe_ld(X + 16);
                            bool got = false;
                            while ((int32_t)(pipe_ld(X + 32) - hd) > 0) {
                                uint32_t seen = hd;
                                if (__hip_atomic_compare_exchange_strong(X + 16, &seen, hd + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { got = true; break; }
                                hd = seen;
                            }
                            if (got) {
                                const unsigned long long* sl = slots + (hd & (A.cap - 1u));
                                const unsigned long long t0 = wall_clock64();
                                for (;;) {
                                    const unsigned long long v = __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    if ((uint32_t)(v >> 32) == hd + 1u) { item = (uint32_t)v; break; }
                                    if (pipe_ld(abort_w)) break;
                                    if (wall_clock64() - t0 > PIPE_SPIN_LIMIT) { pipe_st(abort_w + 28, hd); pipe_st(abort_w + 29, pipe_ld(X + 32)); pipe_st(abort_w + 31, (uint32_t)(wall_clock64() - t0)); pipe_st(abort_w, 3u); break; }
                                    __builtin_amdgcn_s_sleep(1);
                                }
                                if (item != PIPE_EXIT) {
                                    bits = __hip_atomic_load(P.tile_mask + (item >> 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    pipe_acquire();
                                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                }
                            }
                        }
                        sh_word[1] = item;
                        sh_word[2] = (uint32_t)bits;
                        sh_word[3] = (uint32_t)(bits >> 32);
                    }
                    lds_barrier();
                    const uint32_t item = lds_uniform(sh_word[1]);
                    const unsigned long long bits = ((unsigned long long)lds_uniform(sh_word[3]) << 32) | lds_uniform(sh_word[2]);
                    if (item != PIPE_EXIT) {
                        stage_block<NX, false, 256, VAR == 2 ? REF_VM : 0xFFu>(P, n_mult, n_z, stash_rows, (item >> 8) * 64u + (item & 255u) * (uint32_t)P.bx, bits, lds, or_slots, false, nullptr, false);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        lds_barrier();
                        if (t == 0) {
                            pipe_add(stage_done + (item >> 8), 1u);
                            ++n_help;
                        }
                    }
                    lds_barrier();
                }
            }
            if (all_done) break;
        }
        if (t == 0) { atomicAdd(reinterpret_cast<unsigned long long*>(A.ctl + PIPE_STATS) + 0, waited); pipe_add(A.ctl + PIPE_STATS + 11, 1u); if (n_help) pipe_add(A.ctl + PIPE_ABORT + 21, n_help); }   // (word 21: stage items served by helping Riccati workers; word 14 is the fail count of k_egest)
        return;
    }
    // ================================================================ stage worker: pulls (tile, sub-block) items of its XCD
    unsigned long long busy = 0;
    uint32_t n_items = 0;
    bool have_bounds = false;
    for (;;) {
        PIPE_STAMP(11);
        if (t == 0) {
            const uint32_t tk = pipe_add(X + 16, 1u);
            const unsigned long long* sl = slots + (tk & (A.cap - 1u));
            uint32_t item = PIPE_EXIT;
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                const unsigned long long v = __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(v >> 32) == tk + 1u) { item = (uint32_t)v; break; }
                if (pipe_ld(X + 48) >= n_tiles_x || pipe_ld(abort_w)) break;             // every tile of this XCD is finished: nothing can arrive
                if (wall_clock64() - t0 > PIPE_SPIN_LIMIT) { pipe_st(abort_w + 28, tk); pipe_st(abort_w + 29, pipe_ld(X + 32)); pipe_st(abort_w + 31, (uint32_t)(wall_clock64() - t0)); pipe_st(abort_w, 4u); break; }
                __builtin_amdgcn_s_sleep(MPC_PIPE_SLEEP);
            }
            const unsigned long long t1 = wall_clock64();
            PIPE_STAMP(12);
            waited += t1 - t0;
            busy -= t1;
            unsigned long long bits = 0ull;
            if (item != PIPE_EXIT) {
                bits = __hip_atomic_load(P.tile_mask + (item >> 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pipe_acquire();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            sh_word[1] = item;
            sh_word[2] = (uint32_t)bits;
            sh_word[3] = (uint32_t)(bits >> 32);
        }
        lds_barrier();
        const uint32_t item = lds_uniform(sh_word[1]);
        const unsigned long long bits = ((unsigned long long)lds_uniform(sh_word[3]) << 32) | lds_uniform(sh_word[2]);
        if (item == PIPE_EXIT) break;
        PIPE_STAMP(13);
        const uint32_t tile = item >> 8;
        stage_block<NX, false, 256, VAR == 2 ? REF_VM : 0xFFu>(P, n_mult, n_z, stash_rows, tile * 64u + (item & 255u) * (uint32_t)P.bx, bits, lds, or_slots, n_pass == 5u, nullptr, have_bounds);
        // (an item whose instance columns have all finished leaves stage_block before the copy)
        have_bounds = have_bounds || ((bits >> (((item & 255u) * (uint32_t)P.bx) & 63u)) & ((P.bx >= 64) ? ~0ull : ((1ull << P.bx) - 1ull))) != 0ull;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               // this wave's rows are in the L2
        lds_barrier();
        PIPE_STAMP(14);
        if (t == 0) {
            pipe_add(stage_done + tile, 1u);
            busy += wall_clock64();
            ++n_items;
        }
        PIPE_STAMP(15);
        ++n_pass;
    }
#undef PIPE_STAMP
    // ---- hybrid solve: the stage workers of an XCD leave when every tile of the XCD has stopped -- on their way out they put the instances the
    //      tiles left onto the hand-over lists (ho_lists; tiles dealt by ticket).  A launch of its own for this costs 5.5 us between the two kernels.
    if (A.ho_list != nullptr && t < 64 && pipe_ld(abort_w) == 0u) {
        pipe_acquire();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (;;) {
            uint32_t j = 0u;
            if (t == 0) j = pipe_add(X + 56, 1u);
            j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
            if (j >= n_tiles_x) break;
            ho_lists(P, (j * n_xcd + xcd) * 64u + (uint32_t)t, A.ho_list, A.ctl + PIPE_HO, stage_done);
        }
    }
    if (t == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(A.ctl + PIPE_STATS);
        atomicAdd(st + 1, waited);
        atomicAdd(st + 2, busy + wall_clock64());
        atomicAdd(st + 3, (unsigned long long)n_items);
        pipe_add(A.ctl + PIPE_STATS + 10, 1u);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// k_solve_wg: ALL iterations of `bx` instances by ONE workgroup that never lets go of them -- the workgroup-resident solve path.
// k_pipeline chains two latencies per interior-point iteration (a 31-stage one-instance-per-lane Riccati pass of a 64-instance tile,
// 42 us whatever the number of live lanes, and a stage work item, 24 us) plus two hand-offs between workgroups, and a launch lasts
// as many such rounds as its SLOWEST instance needs while most tiles idle through the tail.  Here every workgroup cycles on its own:
//     stage blocks (workspace rows written by phase_finish) -> LDS records
//     wave-per-instance MFMA Riccati (mpc_riccati_mfma.h): backward sweep, forward sweep; P_k / p_k and the step go to the workspace
//     stage_block: the SAME stage phases as every other path (line search, update, derivatives, KKT error, condensed blocks)
// No queue, no hand-off between workgroups, no Riccati role: a workgroup leaves when its last instance is done and the hardware
// dispatcher hands the CU to the next workgroup of the grid.  The rows stay in the tile-major workspace (they are this CU's own
// and come back from the L2), so the stage phases are bit-for-bit those of the other paths; only the KKT solve rounds differently.
// ---------------------------------------------------------------------------------------------------------------
struct PrestartFromWs {
    __device__ __forceinline__ double z(int, int, int) const { return 0.0; }
    __device__ __forceinline__ double ref0(int, int) const { return 0.0; }
};      // prestart_par_block reads the caller's guess and the reference from the workspace rows (k_ingest / a restart wrote them)
template <int NX, class Src = PrestartFromWs> __device__ __forceinline__ void prestart_par_block(const PRef& P, const uint32_t b0, double* sm, const Src src = Src{});

// restart of the instance of a one-instance workgroup of k_solve_wg at a level of its second chance: warm start -> tile-major Z, start-point
// safeguard, start iterate (what k_start does for a block) -- see k_solve_wg<.., RESC>
template <int NX>
__device__ __attribute__((noinline)) void wg_restart(const PRef& Pin, const int n_mult, const int n_z, const int stash_rows, const uint32_t b0, double* lds,
                                                     int (*or_slots)[8], uint32_t* live_out, const bool carry, const bool from_xs, const bool keep_first, const bool first_tiled) {
    const PRef P = mpc_pref(static_cast<const Params&>(Pin), false);      // (a copy in registers, its cache policy a literal again: the reference points into the caller's stack)
    using D = Dim<NX>;
    const int t = threadIdx.x, N = P.N, bx = P.bx;
    struct { int b, k, mb, bl; } c;
    c.k = t / bx;
    c.bl = t & (bx - 1);
    c.mb = (int)blockIdx.x * bx;
    c.b = (int)b0 + c.bl;
    const bool valid = (c.k <= N) && (c.b < P.B);
    if (valid) {
        constexpr int NZ = D::NZ;
        double v[MPC_EV(NZ)];
        static_assert(2 * MPC_EV(NZ) <= D::NPK, "two iterates fit the cost-to-go rows of a stage");
        // (first level: the iterate the first attempt stopped at is kept -- behind the warm start in the same scratch rows --, with its KKT
        //  error; should every level fail, the caller gets that row back, as from rescue_dev: wg_restore_first)
        if (keep_first) {
            // (an instance that came out of the pipeline stalled has never been taken over: its iterate is still tile-major)
            if (first_tiled) ws_load_rows<NZ>(MPC_ROWS(MPC_K(P.Z, NZ, 0, e)), v);
            else ws_load_rows<NZ>(MPC_ROWS(MPC_KI(P.MZ, NZ, 0, e)), v);
            ws_store_rows<NZ>(MPC_ROWS(MPC_K(P.PK, D::NPK, 0, MPC_EV(NZ) + e)), v);
            if (c.k == 0) MPC_S(P.SC, SC_E0S) = (double)MPC_S(P.SC, SC_E0);
        }
        // (the scratch of the warm start: the tile-major rows of the cost-to-go, which this kernel does not use)
        if (carry) {
            ws_load_rows<NZ>(MPC_ROWS(MPC_KI(P.MZ, NZ, 0, e)), v);
            ws_store_rows<NZ>(MPC_ROWS(MPC_K(P.PK, D::NPK, 0, e)), v);
        }
        if (from_xs) {
            if (!carry) ws_load_rows<NZ>(MPC_ROWS(MPC_K(P.PK, D::NPK, 0, e)), v);
        } else {
            const size_t nw = (size_t)2 * N + (size_t)NX * (N + 1);
            const double* xr = P.x0 + (size_t)c.b * nw;
            v[0] = (c.k < N) ? MPC_GP(xr, 2 * c.k) : 0.0;
            v[1] = (c.k < N) ? MPC_GP(xr, 2 * c.k + 1) : 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) v[2 + i] = MPC_GP(xr, 2 * N + NX * c.k + i);
        }
        ws_store_rows<NZ>(MPC_ROWS(MPC_K(P.Z, NZ, 0, e)), v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    prestart_par_block<NX>(P, b0, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage_block<NX, true, 256>(P, n_mult, n_z, stash_rows, b0, ~0ull, lds, or_slots, false, live_out);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// every level of the second chance has failed: the instance goes back to the caller as its FIRST attempt left it (row, status, iteration
// count, KKT error) -- what rescue_dev does by writing back converged rows only
template <int NX>
__device__ __attribute__((noinline)) void wg_restore_first(const PRef& Pin, const uint32_t b0, const int st0, const int it0) {
    const PRef P = mpc_pref(static_cast<const Params&>(Pin), false);
    using D = Dim<NX>;
    constexpr int NZ = D::NZ;
    const int t = threadIdx.x, bx = P.bx;
    struct { int b, k, mb, bl; } c;
    c.k = t / bx;
    c.bl = t & (bx - 1);
    c.mb = (int)blockIdx.x * bx;
    c.b = (int)b0 + c.bl;
    if (c.k <= P.N && c.b < P.B) {
        double v[MPC_EV(NZ)];
        ws_load_rows<NZ>(MPC_ROWS(MPC_K(P.PK, D::NPK, 0, MPC_EV(NZ) + e)), v);
        ws_store_rows<NZ>(MPC_ROWS(MPC_KI(P.MZ, NZ, 0, e)), v);          // (the hand-back at the end of the kernel copies these rows ...
        ws_store_rows<NZ>(MPC_ROWS(MPC_K(P.Z, NZ, 0, e)), v);             //  ... unless no level ever got a round: then these are what k_egest reads)
        if (c.k == 0) {
            MPC_S(P.SC, SC_E0) = (double)MPC_S(P.SC, SC_E0S);
            MPC_S(P.ISC, IS_STATUS) = st0;
            MPC_S(P.ISC, IS_ITERS) = it0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// RESC: the SECOND CHANCE of an instance that stalls (status 0 / -7: what rescue_dev does on the host, see there) starts inside the running
// launch, in the wavefront that owns the instance -- one instance per workgroup (bx = 1), so the level's lower bound of the circle rows and
// its tolerance are the workgroup's own copies of P.ol / P.tol.  An instance that stalls at iteration 4 ... 32 does not wait for the
// slowest of the batch before it is looked at again, and its levels need no launches, no compaction and no host round trip.  Schedule and
// semantics are rescue_dev's: pass 1 = levels {0, 1} x the true bound, pass 2 = {0, 0.4, 0.7, 0.9, 1}; a level starts from the last
// level of its pass that converged (else from the caller's x0); levels below the last stop at 1e-4; the last level IS the original NLP.
struct WgRescue { double ol_raw, relax; int32_t on; };
constexpr int RESC_LEVELS = 8;          // index q of IS_RLEV: 0 = the first attempt, 1-2 pass 1, 3-7 pass 2
__device__ __forceinline__ double resc_fraction(int q) { return q == 2 || q == 7 ? 1.0 : q == 4 ? 0.4 : q == 5 ? 0.7 : q == 6 ? 0.9 : 0.0; }
__device__ __forceinline__ bool resc_last(int q) { return q == 2 || q == 7; }

// LDS of k_solve_wg, in doubles: the pad record (also the dump area of the sweeps: one double per lane), the records, the bounds table
template <int NX>
struct WgLds {
    static constexpr int PAD = Rec<NX>::SIZE > 64 ? Rec<NX>::SIZE : 64;
    __host__ __device__ static constexpr size_t doubles(int S, int bxw) { return (size_t)PAD + (size_t)Rec<NX>::SIZE * S * bxw + (size_t)2 * S * (NX + 2) + (size_t)16 * bxw + (size_t)64 * NX + (size_t)WgScl::SIZE * bxw; }
};
// k_solve_wg: the caller's output pointers (and the pointers of its epilogue) are read from the kernel-argument segment WHERE THEY ARE USED -- behind
// an opaque copy of the segment pointer, so that the compiler does not load them at the kernel's entry and carry ten scalar registers through the
// rounds (the stage phases run at 450+ vector registers with ~400 scalar values spilled into lanes; with the pointers live the kernel went from 456
// to 469 registers and onto scratch).  Params is the kernel's first argument: offset 0 of the segment.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) Params* mpc_kernarg_params;
__device__ __forceinline__ mpc_kernarg_params wg_kernarg() {
    mpc_kernarg_params kp = (mpc_kernarg_params)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}
__device__ __forceinline__ EmitDst wg_emit_dst() {
    const mpc_kernarg_params kp = wg_kernarg();
    return EmitDst{kp->x_out, kp->status_out, kp->iters_out, kp->kkt_out, kp->fail_count};
}
#else
inline EmitDst wg_emit_dst() { return EmitDst{}; }
#endif
template <int NX, int VAR, bool RESC>
__device__ __forceinline__ void solve_wg_body(const Params& Pk, const int n_mult, const int n_z, const int stash_rows, uint32_t* stats, const uint32_t* skip_if,
                                              const WgRescue& resc, unsigned long long* wtrace, const int32_t* list, const uint32_t* list_n) {
    PRef P(Pk);                                   // (RESC: ol and tol of the level an instance is at)
#if defined(__HIP_DEVICE_COMPILE__)
    // RESC + emit: the row that counts is the one the instance ends with after all its levels (an intermediate level converges to a relaxed problem,
    // a failed pass gives the first attempt's row back): written once, when the workgroup leaves -- not by the phases
    const bool emit_exit = RESC && resc.on && Pk.emit != 0;
    if (RESC && emit_exit) P.emit = 0;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int or_slots[2][8];
    __shared__ uint32_t sh_mask;
    __shared__ int sh_next[2];
    if (skip_if != nullptr && *skip_if != 0u) return;          // the pipeline launch in front of this one was abandoned: the host starts over
    const double tol0 = P.tol;
    bool fresh = true;                            // the instance's rows are tile-major (start iterate / hand-over from the pipeline): take them over
    using D = Dim<NX>;
    using RC = Rec<NX>;
    const int bx = P.bx, t = threadIdx.x, N = P.N;
    const int lane = t;                          // ONE wavefront per workgroup (bx = 1 or 2 instances, (N + 1) * bx <= 64 stage threads)
    const uint32_t b0 = (blockIdx.x + (uint32_t)P.tile0 * (64u / (uint32_t)bx)) * (uint32_t)bx;
    // The instances of this workgroup (slots 0 and 1 of its block; -1: none).  Without a list: b0 and its neighbour.  Behind the pipeline: from the
    // list of instances its retiring tiles have left -- workgroup i takes entries i and i + gridDim.x, so that the launch needs no more
    // workgroups than the machine holds at once and every one of them starts at once (with two instances per workgroup by position, 1 163 of the
    // 2 048 workgroups of the headline batch had work, 139 more than there are slots: the last of them started 90 us into a 250 us launch).
    // The lists are one per bucket of the KKT error at the hand-over, largest first: V = their concatenation, longest-running instances first.
    // Workgroup i takes V[i]; the n - G entries beyond the G workgroups go, shortest last, to the workgroups from G - 1 downwards -- the
    // instances with the most iterations in front of them have a wavefront to themselves (a round with one live instance takes 25 us, with two 34).
    int ib0 = (int)b0 < P.B ? (int)b0 : -1, ib1 = (bx > 1 && (int)b0 + 1 < P.B) ? (int)b0 + 1 : -1;
    uint32_t in_mb = 0u;
    if (list != nullptr) {
        uint32_t cnt[HO_BUCKETS], n = 0u;
#pragma unroll
        for (int q = 0; q < (int)HO_BUCKETS; ++q) { cnt[q] = list_n[q]; n += cnt[q]; }
        auto entry = [&](uint32_t v) -> int {
            if (v >= n) return -1;
#pragma unroll
            for (int q = 0; q < (int)HO_BUCKETS; ++q) { if (v < cnt[q]) return list[(uint32_t)q * (uint32_t)P.Bp + v]; v -= cnt[q]; }
            return -1;
        };
        const uint32_t G = gridDim.x;
        // (list entries: loaded with vector instructions although every lane asks for the same word -- uniform values for the whole launch: scalar registers)
        ib0 = lds_uniform(entry(blockIdx.x));
        ib1 = bx > 1 ? lds_uniform(entry(2u * G - 1u - blockIdx.x)) : -1;            // (V[G + m] belongs to workgroup G - 1 - m)
        if (ib0 < 0) return;
        // (bit 30 of an entry: the pipeline's last stage items of the instance's tile wrote the mailbox arrays -- nothing to copy at the takeover)
        in_mb = ((ib0 & HO_IN_MB) ? 1u : 0u) | ((ib1 >= 0 && (ib1 & HO_IN_MB)) ? 2u : 0u);
        ib0 &= ~HO_IN_MB;
        if (ib1 >= 0) ib1 &= ~HO_IN_MB;
    }
    struct { int b, k, mb, bl; } c;              // (the workspace accessors are written in terms of c.b / c.k; MPC_KI: c.mb / c.bl)
    c.k = t / bx;
    c.bl = t & (bx - 1);
    c.mb = (int)blockIdx.x * bx;
    c.b = c.bl ? ib1 : ib0;
    const bool valid = (c.k <= N) && (c.b >= 0) && (c.b < P.B);
    // LDS: [one pad record -- what the sweeps' operand prefetch of "stage -1" reads and what their lanes without an entry write | records
    //       of the workgroup's instances, Rec<NX> | bounds table of the stage phases]; nothing else: one wavefront needs no reduction
    //       scratch and exchanges neighbour stages by lane shuffles
    const mpc_lds_ptr recs = (mpc_lds_ptr)(lds_ptr_t)lds + WgLds<NX>::PAD;
    const mpc_lds_ptr dump = (mpc_lds_ptr)(lds_ptr_t)lds;
    double* const lds_bnd = lds + WgLds<NX>::PAD + (size_t)RC::SIZE * ((N + 1) * bx);
    double* const lds_c0 = lds_bnd + 2 * (N + 1) * D::NZ;       // 8 doubles per instance: c_0 of the last stage phases
    double* const lds_r0 = lds_c0 + 8 * bx;                     // 8 per instance: r_0
    double* const lds_rn = lds_r0 + 8 * bx;                     // NX per stage thread: the reference of its next stage
    double* const lds_scl = lds_rn + 64 * NX;                   // WgScl::SIZE per instance: scalars and filter
    bool bounds_ok = false;
    // per instance (slot g of the workgroup), wave-uniform and in registers from round to round: the last inertia correction, the mark of
    // heavily weighted circle rows -- read from the workspace when the instances are taken over
    double dl0 = 0.0, dl1 = 0.0;
    WgIo io{0u, 0u, lds_c0, lds_rn, lds_r0, lds_scl, true, false};
    const MfmaWords lane_words = mfma_lane_load<NX>(lane);
#define WG_STAMP(i) do { if (P.DBG && t == 0 && rounds == 3u) P.DBG[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    uint32_t rounds = 0, sweeps = 0, inst_rounds = 0, prev_mask = 0u;
    const unsigned long long t_begin = wtrace ? wall_clock64() : 0ull;        // (option wg_trace: when did this workgroup start, how long did it run, how many rounds)
    unsigned long long t_fresh = 0ull, t_round1 = 0ull;                        // (... when were its instances taken over, when was its first round done)
    if (RESC && resc.on && t == 0 && ib0 >= 0 && ib0 < P.B) MPC_UB(P.ISC, (uint32_t)IS_RLEV, ib0) = 0;       // the first attempt (ib0: with one instance per workgroup, THE instance -- b0 only without a list)
    for (;;) {
        // ---- which of my instances are iterating: the status rows of the workspace in the first round; after that stage_block has
        //      left the mask in sh_mask (an instance the sweeps gave up on is inactive there: phase_load_scalars reads its status)
        if (rounds == 0u && t < 64) {
            const int bb = t ? ib1 : ib0;
            // (the hand-over lists hold instances that were iterating when their tile left the pipeline: no look at the status row)
            const bool run = t < bx && bb >= 0 && bb < P.B && (list != nullptr || (int32_t)MPC_UB(P.ISC, (uint32_t)IS_STATUS, bb >= 0 ? bb : 0) == ST_RUNNING);
            const unsigned long long mk = __ballot(run ? 1 : 0);
            if (t == 0) sh_mask = (uint32_t)mk;
        }
        lds_barrier();
        const uint32_t mask = lds_uniform(sh_mask);
        // ---- launches that write the caller's rows themselves (Params::emit): an instance that has just left the mask -- converged, out of iterations,
        //      stalled, given up by the sweeps -- goes out now, from the rows and scalars its last round left in memory (this wavefront's own stores,
        //      drained at the end of that round).  Here and not in the phases: they run at 450+ registers, this is six loads and stores per lane.
        if (Pk.emit != 0 && !emit_exit) {
            const uint32_t gone = prev_mask & ~mask;
            if (gone != 0u && valid && ((gone >> c.bl) & 1u)) {
                double z[MPC_EV(D::NZ)];
                ws_load_rows<D::NZ>(MPC_ROWS(MPC_KI(P.MZ, D::NZ, 0, e)), z);
                int st = 0, it = 0;
                double e0 = 0.0;
                if (c.k == 0) { st = (int32_t)MPC_S(P.ISC, IS_STATUS); it = (int32_t)MPC_S(P.ISC, IS_ITERS); e0 = MPC_S(P.SC, SC_E0); }
                emit_row<NX>(wg_emit_dst(), N, c.b, c.k, z, st, it, e0);
            }
            prev_mask = mask;
        }
        if (mask == 0u) {
            if (!RESC || !resc.on) break;
            // ---- the instance of this workgroup has stopped: is a (further) level of the second chance due?
            if (t == 0) {
                const int bb = ib0;
                int next = -1, carry = 0;
                if (bb >= 0 && bb < P.B) {
                    const int st = (int32_t)MPC_UB(P.ISC, (uint32_t)IS_STATUS, bb), it = (int32_t)MPC_UB(P.ISC, (uint32_t)IS_ITERS, bb);
                    const int lev = (int32_t)MPC_UB(P.ISC, (uint32_t)IS_RLEV, bb);
                    const int q = lev & 0xFF;
                    int has_xs = (lev >> 8) & 1, first = lev & ~0x1FF;                  // (first: status and iteration count of the first attempt)
                    int acc = (q == 0 ? 0 : (int32_t)MPC_UB(P.ISC, (uint32_t)IS_ITACC, bb)) + it;
                    if (q == 0) {
                        if (st == 0 || st == -7) { next = 1; first = (st == -7 ? 0x200 : 0) | (it << 16); if (stats != nullptr) atomicAdd(stats + 4, 1u); }
                    } else {
                        if (st == 1) { carry = 1; has_xs = 1; }
                        if (!resc_last(q)) next = q + 1;
                        // pass 2 starts over from the caller's x0; the levels of a pass that ended open are not counted (rescue_dev: k_rescue_gather)
                        else if (st != 1 && q == 2) { next = 3; has_xs = 0; acc = first >> 16; }
                    }
                    if (next >= 0) {
                        MPC_UB(P.ISC, (uint32_t)IS_RLEV, bb) = next | (has_xs << 8) | first;
                        MPC_UB(P.ISC, (uint32_t)IS_ITACC, bb) = acc;
                        next |= has_xs << 8;
                    } else if (q != 0) {
                        if (st == 1) MPC_UB(P.ISC, (uint32_t)IS_ITERS, bb) = acc;      // (what the caller is told: the first attempt and the levels of the pass that brought it in)
                        else next = -2 - ((first >> 9) & 1) - 2 * (first >> 16);       // every level failed: -2 - (first status was -7) - 2 * (its iterations)
                    }
                }
                sh_next[0] = next;
                sh_next[1] = carry;
            }
            lds_barrier();
            const int nx_lev = lds_uniform(sh_next[0]), carry = lds_uniform(sh_next[1]);
            lds_barrier();
            if (nx_lev <= -2) {
                const int code = -2 - nx_lev;
                const PRef Pc = P;
                wg_restore_first<NX>(Pc, (uint32_t)ib0, (code & 1) ? -7 : 0, code >> 1);
            }
            if (nx_lev < 0) break;
            const int q = nx_lev & 0xFF;
            const bool from_xs = (nx_lev >> 8) != 0;
            // the level's problem: lower bound of the circle rows (relaxed like mpc_set_bounds relaxes it), tolerance
            {
                const double lo = resc_fraction(q) * resc.ol_raw;
                P.ol = lo - resc.relax * fmax(1.0, fabs(lo));
                P.tol = resc_last(q) ? tol0 : fmax(tol0, 1e-4);
            }
            {
                // (out of line, on a copy of the parameters: the start-point safeguard and the start iterate inlined here put the hot loop at 512
                //  registers with scratch -- B = 256 lane following paid 1.7 % for a path it never takes)
                const PRef Pc = P;
                wg_restart<NX>(Pc, n_mult, n_z, stash_rows, (uint32_t)ib0, lds, or_slots, &sh_mask, carry != 0, from_xs, q == 1, rounds == 0u);
            }
            fresh = true;
            bounds_ok = false;
            io.have = false;                    // (the restart used the whole LDS)
            continue;
        }
        ++rounds;
        inst_rounds += (uint32_t)__popc(mask);
        WG_STAMP(12);
        // ---- taking the instances over: the iterate, its multipliers and the reference move from the tile-major arrays (where the
        //      pipeline / the start-iterate kernel left them) into the instance-major mailbox arrays the rounds below work on -- one
        //      wavefront reads all stages of its one or two instances, and only there are the pieces of a thread contiguous
        // (not for the instances of a tile whose last stage items in the pipeline wrote the mailbox arrays themselves: in_mb, from the list entry)
        const bool by_pipe = !RESC && valid && ((in_mb >> c.bl) & 1u) != 0u;
        if (fresh && valid && !by_pipe) {
            auto move = [&](auto cnt, auto from, auto to) {
                constexpr int CNT = decltype(cnt)::value;
                double v[MPC_EV(CNT)];
                ws_load_rows<CNT>(from, v);
                ws_store_rows<CNT>(to, v);
            };
            constexpr int NZ = D::NZ;
            move(std::integral_constant<int, NZ>{}, MPC_ROWS(MPC_K(P.Z, NZ, 0, e)), MPC_ROWS(MPC_KI(P.MZ, NZ, 0, e)));
            move(std::integral_constant<int, NZ>{}, MPC_ROWS(MPC_K(P.ZL, NZ, 0, e)), MPC_ROWS(MPC_KI(P.MZL, NZ, 0, e)));
            move(std::integral_constant<int, NZ>{}, MPC_ROWS(MPC_K(P.ZU, NZ, 0, e)), MPC_ROWS(MPC_KI(P.MZU, NZ, 0, e)));
            move(std::integral_constant<int, 4>{}, MPC_ROWS(MPC_K(P.SO, 3, 0, e)), MPC_ROWS(MPC_KI(P.MSO, 3, 0, e)));
            move(std::integral_constant<int, 4>{}, MPC_ROWS(MPC_K(P.NUO, 3, 0, e)), MPC_ROWS(MPC_KI(P.MNUO, 3, 0, e)));
            move(std::integral_constant<int, 4>{}, MPC_ROWS(MPC_K(P.ZLO, 3, 0, e)), MPC_ROWS(MPC_KI(P.MZLO, 3, 0, e)));
            move(std::integral_constant<int, 4>{}, MPC_ROWS(MPC_K(P.ZUO, 3, 0, e)), MPC_ROWS(MPC_KI(P.MZUO, 3, 0, e)));
            move(std::integral_constant<int, NX>{}, MPC_ROWS(MPC_K(P.LAM, NX, 0, e)), MPC_ROWS(MPC_KI(P.MLAM, NX, 0, e)));
        }
        if (fresh) {
            // (the rows just moved are read by OTHER lanes too -- the neighbour stage's --: they are in the L2 before anything loads them)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // ---- stage blocks -> LDS records (every stage thread its own; defect negated, three constants, Hux of stage 0).  Only when the
        //      instances have just been taken over (the blocks were written tile-major by the start-iterate kernel or the pipeline) and
        //      when a sweep has to be repeated (from the copy in the mailbox): the rounds' own blocks go from the stage phases straight
        //      into the records.
        const bool whole = fresh, tiled = fresh && !by_pipe;
        auto build_records = [&]() {
            if (valid && ((mask >> (t & (bx - 1))) & 1u)) {
                double blk[MPC_EV(D::NBLK)];
                if (tiled) ws_load_rows<D::NBLK>(MPC_ROWS(MPC_K(P.BLK, D::NBLK, 0, e)), blk);
                else ws_load_rows<D::NBLK>(MPC_ROWS(MPC_KI(P.MBLK, D::NBLK, 0, e)), blk);
                double hx0 = 0.0, hx1 = 0.0;
                if (c.k == 0) { hx0 = MPC_S(P.SC, SC_HUX0); hx1 = MPC_S(P.SC, SC_HUX1); }
                const mpc_lds_ptr r = recs + ((t & (bx - 1)) * (N + 1) + c.k) * RC::SIZE;
                // (a rebuild from the mailbox copy restores what the cost-to-go went over; A and the defect never left the record -- and are not in the copy)
#pragma unroll
                for (int i = 0; i < D::NBLK; ++i) {
                    const bool kept = i < D::B_RUU || (i >= D::B_CN && i < D::B_CN + NX);
                    if (whole || !kept) r[RC::slot(i)] = (i >= D::B_CN && i < D::B_CN + NX) ? -blk[i] : blk[i];
                }
                r[RC::ZERO] = 0.0;
                r[RC::ONE] = 1.0;
                r[RC::DT] = P.dt;
                r[RC::HX] = hx0;
                r[RC::HX + 1] = hx1;
            }
        };
        if (fresh) {
            build_records();
            const int a0 = ib0 >= 0 ? ib0 : 0, a1 = ib1 >= 0 ? ib1 : a0;
            dl0 = lds_uniform((double)MPC_UB(P.SC, (uint32_t)SC_DLAST, a0));
            dl1 = lds_uniform((double)MPC_UB(P.SC, (uint32_t)SC_DLAST, a1));
            io.ill = ((int32_t)MPC_UB(P.ISC, (uint32_t)IS_ILL, a0) != 0 ? 1u : 0u) | ((bx > 1 && (int32_t)MPC_UB(P.ISC, (uint32_t)IS_ILL, a1) != 0) ? 2u : 0u);
            if (valid && c.k == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i) lds_c0[(t & (bx - 1)) * 8 + i] = MPC_S(P.SC, SC_C0 + i);
                // the scalars the phases read every round (WgScl)
                double* q = lds_scl + c.bl * WgScl::SIZE;
                q[WgScl::MU] = MPC_S(P.SC, SC_MU); q[WgScl::TAU] = MPC_S(P.SC, SC_TAU); q[WgScl::DF] = MPC_S(P.SC, SC_DF); q[WgScl::THETA] = MPC_S(P.SC, SC_THETA);
                q[WgScl::FCOST] = MPC_S(P.SC, SC_FCOST); q[WgScl::LOGSUM] = MPC_S(P.SC, SC_LOGSUM); q[WgScl::THMAX] = MPC_S(P.SC, SC_THMAX);
                q[WgScl::THMIN] = MPC_S(P.SC, SC_THMIN); q[WgScl::A0LB] = MPC_S(P.SC, SC_A0LB); q[WgScl::A0UB] = MPC_S(P.SC, SC_A0UB);
                q[WgScl::STATUS] = (double)(int32_t)MPC_S(P.ISC, IS_STATUS); q[WgScl::NFILT] = (double)(int32_t)MPC_S(P.ISC, IS_NFILT);
                q[WgScl::ITERS] = (double)(int32_t)MPC_S(P.ISC, IS_ITERS); q[WgScl::CONV] = (double)(int32_t)MPC_S(P.ISC, IS_CONV);
                q[WgScl::FROW] = (double)(int32_t)MPC_S(P.ISC, IS_FROW); q[WgScl::HAVETH0] = (double)(int32_t)MPC_S(P.ISC, IS_HAVETH0);
            }
            // ... and the filter: lane l fetches row l of each instance
            {
                const bool f0 = ib0 >= 0 && t < 2 * FILTER_MAX, f1 = bx > 1 && ib1 >= 0 && t < 2 * FILTER_MAX;
                const double v0 = f0 ? (double)ws_ref3(P, P.FILT, 0u, (uint32_t)ib0, mpc_prow((uint32_t)t)) : 0.0;
                const double v1 = f1 ? (double)ws_ref3(P, P.FILT, 0u, (uint32_t)ib1, mpc_prow((uint32_t)t)) : 0.0;
                if (f0) lds_scl[WgScl::FILT + t] = v0;
                if (f1) lds_scl[WgScl::SIZE + WgScl::FILT + t] = v1;
            }
        }
        if (!bounds_ok) {
            const int nb = (N + 1) * D::NZ;
            fill_bounds<8>(P, lds_bnd, nb, t, 64);
            bounds_ok = true;
        }
        lds_barrier();
        if (wtrace != nullptr && fresh && t_fresh == 0ull) t_fresh = wall_clock64() - t_begin;
        fresh = false;
        WG_STAMP(13);
        // (per-lane operand offsets of the sweeps: unpacked every round from the lane's four table words -- opaque to the compiler here, so
        //  that the twenty unpacked values are not hoisted out of the loop and kept in registers across stage_block)
        MfmaWords lw = lane_words;
        asm volatile("" : "+v"(lw.w[0]), "+v"(lw.w[1]), "+v"(lw.w[2]), "+v"(lw.w[3]));
        MfmaLane<NX> m;
        mfma_lane_setup<NX>(m, lw, P.dt);
        // ---- KKT solves of the live instances (two: interleaved in one instruction stream), then the stage phases (wg_stage)
        constexpr uint32_t SVM = VAR == 2 ? REF_VM : 0xFFu;
        {
            auto inst_of = [&](int g, MfmaInst& in, double& x0) {
                in.inst = (uint32_t)(g ? ib1 : ib0);
                in.delta_last = g ? dl1 : dl0;
                in.sym_hint = ((io.ill >> g) & 1u) != 0u;
                // x~_0 = (-c_0, 0.., 1) as B operand of the forward sweep (requested now, needed after the backward sweep)
                x0 = 0.0;
                if ((lane & 3) == 0) {
                    if (m.Rb < NX) x0 = -lds_c0[g * 8 + m.Rb];
                    else if (m.Rb == 7) x0 = 1.0;
                }
            };
            io.fail = 0u;
            auto finish = [&](int g, bool ok, double delta) {
                const int bb = g ? ib1 : ib0;
                if (ok) { if (delta > 0.0) { if (g) dl1 = lds_uniform(delta); else dl0 = lds_uniform(delta); } }
                else io.fail |= 1u << g;
                if (lane != 0) return;
                // (the workspace rows follow for whoever looks at the instance after this launch; nothing of this round reads them)
                if (ok) {
                    if (delta > 0.0) MPC_UB(P.SC, (uint32_t)SC_DLAST, bb) = delta;
                    MPC_UB(P.SC, (uint32_t)SC_DELTA, bb) = delta;
                } else {
                    MPC_UB(P.ISC, (uint32_t)IS_STATUS, bb) = -7;
                }
            };
            // a sweep that is repeated with an inertia correction starts from the stage blocks again: its cost-to-go went over them
            auto rebuild = [&]() { lds_barrier(); build_records(); lds_barrier(); };
            const int g0 = __builtin_ctz(mask);
            const bool two = (mask & (mask - 1u)) != 0u;
            const int g1 = two ? 1 : g0;
            MfmaInst in[2];
            double x0[2], delta[2];
            bool ok[2];
            inst_of(g0, in[0], x0[0]);
            inst_of(g1, in[1], x0[1]);
            const mpc_lds_ptr rec[2] = {recs + g0 * (N + 1) * RC::SIZE, recs + g1 * (N + 1) * RC::SIZE};
            typedef const MfmaInst (&In1)[1];
            typedef const mpc_lds_ptr (&Rec1)[1];
            typedef double (&D1)[1];
            typedef const double (&CD1)[1];
            typedef bool (&B1)[1];
            typedef const bool (&CB1)[1];
            if (two) mfma_backward<NX, 2>(P, m, in, rec, lane, dump, delta, ok, sweeps, rebuild);
            else mfma_backward<NX, 1>(P, m, reinterpret_cast<In1>(in), reinterpret_cast<Rec1>(rec), lane, dump, reinterpret_cast<D1>(delta), reinterpret_cast<B1>(ok), sweeps, rebuild);
            if (t == 0) sh_mask = 0u;                                 // (the stage phases leave early, before their ballot, when nothing is active)
            // ---- the stage work of the round, with the forward sweep laid between its loads from memory and its reads of the records
            wg_stage<NX, SVM>(P, ib0, ib1, lds_bnd, recs, n_mult, n_z, rounds == 3u, &sh_mask, io, [&]() {
                if (two) mfma_forward<NX, 2>(P, m, in, rec, lane, dump, x0, ok);
                else mfma_forward<NX, 1>(P, m, reinterpret_cast<In1>(in), reinterpret_cast<Rec1>(rec), lane, dump, reinterpret_cast<CD1>(x0), reinterpret_cast<CB1>(ok));
                finish(g0, ok[0], delta[0]);
                if (two) finish(g1, ok[1], delta[1]);
                lds_barrier();                                       // (step and cost-to-go are in the records; nothing of the sweeps' went to memory that this round reads)
                WG_STAMP(14);
            });
            io.have = true;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's rows are in the L2
        lds_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WG_STAMP(15);
        if (wtrace != nullptr && rounds == 1u) t_round1 = wall_clock64() - t_begin;
    }
#undef WG_STAMP
    // the iterate goes back to the tile-major rows k_egest reads (a workgroup that found nothing to do never moved it) -- unless the launch writes
    // the caller's rows itself (Params::emit: the phases have, at the moment the status became final)
    if (rounds > 0u && valid && !Pk.emit) {
        double v[D::NZ];
        ws_load_rows<D::NZ>(MPC_ROWS(MPC_KI(P.MZ, D::NZ, 0, e)), v);
        ws_store_rows<D::NZ>(MPC_ROWS(MPC_K(P.Z, D::NZ, 0, e)), v);
    }
    if (RESC && emit_exit && valid) {
        // (one instance per workgroup: what it ends with -- the last level's row, or the first attempt's that wg_restore_first put back)
        double z[MPC_EV(D::NZ)];
        if (rounds > 0u) ws_load_rows<D::NZ>(MPC_ROWS(MPC_KI(P.MZ, D::NZ, 0, e)), z);
        else ws_load_rows<D::NZ>(MPC_ROWS(MPC_K(P.Z, D::NZ, 0, e)), z);
        int st = 0, it = 0;
        double e0 = 0.0;
        if (c.k == 0) { st = (int32_t)MPC_S(P.ISC, IS_STATUS); it = (int32_t)MPC_S(P.ISC, IS_ITERS); e0 = MPC_S(P.SC, SC_E0); if (st == ST_RUNNING) st = 0; }
        emit_row<NX>(wg_emit_dst(), N, c.b, c.k, z, st, it, e0);
    }
    if (wtrace != nullptr && t == 0) {
        wtrace[blockIdx.x * 4 + 0] = t_begin;
        wtrace[blockIdx.x * 4 + 1] = wall_clock64();
        wtrace[blockIdx.x * 4 + 2] = rounds;
        wtrace[blockIdx.x * 4 + 3] = (unsigned long long)(inst_rounds & 0xFFFFu) | ((t_fresh & 0xFFFFFFull) << 16) | ((t_round1 & 0xFFFFFFull) << 40);
    }
    if (stats != nullptr && t == 0) {
        atomicMax(stats + 0, rounds);
        atomicAdd(stats + 1, rounds);
        atomicAdd(stats + 2, sweeps);
        atomicAdd(stats + 3, inst_rounds);
    }
#endif
}
// Params::fin_ctl: the launch is the last of its solve -- the workgroup that leaves LAST copies the solve's statistics block (abort word, rounds,
// counters of both loop kernels, instances that did not converge: PIPE_FIN_WORDS words at fin_ctl) into the handle's pinned host block (fin_host),
// so that the host reads it when the stream has drained: no copy command, no kernel of its own behind the loop.  fin_ctl[PIPE_FIN_TICKET] counts the
// workgroups that have left.
constexpr uint32_t PIPE_FIN_WORDS = 32, PIPE_FIN_TICKET = 22;
constexpr int PIPE_MAX_TILES = 256;          // tiles of one persistent launch in converged mode (a Riccati worker owns up to 32: its `fin` mask)
template <int NX, int VAR, bool RESC = false>
__global__ void __launch_bounds__(256) k_solve_wg(const Params Pk, const int n_mult, const int n_z, const int stash_rows, uint32_t* stats, const uint32_t* skip_if,
                                                                   const WgRescue resc, unsigned long long* wtrace, const int32_t* list, const uint32_t* list_n) {
    solve_wg_body<NX, VAR, RESC>(Pk, n_mult, n_z, stash_rows, stats, skip_if, resc, wtrace, list, list_n);
#if defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x < 64) {           // (one wavefront per workgroup)
        const mpc_kernarg_params kp = wg_kernarg();
        uint32_t* const ctl = kp->fin_ctl;
        if (ctl == nullptr) return;
        uint32_t tk = 0u;
        if (threadIdx.x == 0) {
            // (this workgroup's counters are agent-scope atomics: performed once acknowledged -- no cache write-back needed for the last workgroup to
            //  read them with agent-scope loads; the caller's rows become visible with the end of the kernel)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tk = __hip_atomic_fetch_add(ctl + PIPE_FIN_TICKET, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // counted as gone
        }
        tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
        if (tk == gridDim.x - 1u && threadIdx.x < PIPE_FIN_WORDS) {
            const uint32_t v = __hip_atomic_load(ctl + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(kp->fin_host + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
#endif
}

// Start-point safeguard with one thread per (instance, stage), the layout of the stage kernel (bx instance columns per workgroup).  The two
// 30-stage chains of prestart_chain (mpc_stage_math.h: what the CPU harness steps) are chains of sin / cos / tan only because of how they are written: the defect of
// the caller's state guess needs no recursion at all, and in the rollout (delta, v) depend on the controls alone, psi on (delta, v)
// and (x, y, s) on (v, psi) -- so every transcendental is evaluated by the thread of its stage, and what is left sequential are
// three short scans x_{k+1} = push_in(x_k + dt f_k) per state, run by the first two stage-threads of every instance from increments
// parked in LDS.  Same arithmetic per element and the same left-to-right order of the defect sums as prestart_chain.
// Src: where the raw guess z(k, i) (i < 2: input, else state i - 2 of stage k) and the reference of stage 0 come from -- the workspace rows, or
// (k_start) the block's rows of the caller's buffers as they lie in LDS: then nothing here waits for the ingest's stores
// LDS doubles of prestart_par_block: bounds table and its pushed limits [4][S][NZ], rollout / defects / increments [(3 NX + 2)][S][bx], 4 per instance
__host__ __device__ constexpr size_t prestart_doubles(int NX, int S, int bx) { return (size_t)4 * S * (NX + 2) + (size_t)(3 * NX + 2) * S * bx + (size_t)4 * bx; }
template <int NX, class Src>
__device__ __forceinline__ void prestart_par_block(const PRef& P, const uint32_t b0, double* sm, const Src src) {
    constexpr bool WS = std::is_same<Src, PrestartFromWs>::value;
    constexpr int NZ = NX + 2;
    const int N = P.N, S = N + 1, bx = P.bx, t = threadIdx.x, bl = t & (bx - 1);
    const int nb = S * NZ, SB = S * bx;
    double* LBt = sm;                       // [S][NZ]
    double* UBt = LBt + nb;
    double* XR = UBt + nb;                  // [NX][S][bx]  rollout states
    double* DR = XR + NX * SB;              // [NX][S][bx]  rollout: clipping defect of state i at stage k
    double* DG = DR + NX * SB;              // [NX][S][bx]  guess: dynamics defect of state i at stage k
    double* IN = DG + NX * SB;              // [2][S][bx]   increments f_i of the two states being scanned
    double* A0 = IN + 2 * SB;               // [4][bx]      a0lb, a0ub, defect of the guess, 'a rolled-out state was clipped'
    double* PLt = A0 + 4 * bx;              // [S][NZ]      the bounds pushed inwards as push_in does it: what the scans clip against
    double* PHt = PLt + nb;
    struct { int b, k; } c;
    c.k = t / bx;
    c.b = (int)b0 + bl;
    const bool valid = c.k <= N && c.b < P.B;
    const uint32_t bb = (uint32_t)c.b;
#define PS_STAMP(i) do { if (P.DBG && threadIdx.x == 0) P.DBG[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    // the bounds table and, next to it, the bounds pushed inwards the way push_in does it (push_limits) -- every clip below is then fmax / fmin
    for (int q0 = 0; q0 < nb; q0 += 2 * (int)blockDim.x) {
        double lo[2], hi[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int q = q0 + j * (int)blockDim.x + t; lo[j] = q < nb ? (double)MPC_GP(P.LB, q) : 0.0; hi[j] = q < nb ? (double)MPC_GP(P.UB, q) : 0.0; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + j * (int)blockDim.x + t;
            if (q < nb) { double L, H; push_limits(lo[j], hi[j], L, H); LBt[q] = lo[j]; UBt[q] = hi[j]; PLt[q] = L; PHt[q] = H; }
        }
    }
    double a0lb = 0.0, a0ub = 0.0;
    int frow = 1;
    if (valid && c.k == 0) {
        if (WS) frow = prestart_a0<NX>(P, c.b, a0lb, a0ub);
        else frow = prestart_a0_of<NX>(P, src.ref0(bl, 2), src.ref0(bl, 3), a0lb, a0ub);
        A0[bl] = a0lb;
        A0[bx + bl] = a0ub;
    }
    __syncthreads();
    PS_STAMP(1);
    const double dt = P.dt;
#define PP_AT(arr, i, k) (arr)[((i) * S + (k)) * bx + bl]
    if (valid) {
        const int k = c.k;
        double g[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) g[i] = fmin(fmax(WS ? (double)MPC_K(P.Z, NZ, 0, 2 + i) : src.z(bl, k, 2 + i), PLt[k * NZ + 2 + i]), PHt[k * NZ + 2 + i]);
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const double r0 = WS ? (double)MPC_K(P.REF, NX, 0, i) : src.ref0(bl, i);
                const double x0 = fmin(fmax(r0, PLt[2 + i]), PHt[2 + i]);
                PP_AT(XR, i, 0) = x0;
                PP_AT(DR, i, 0) = fabs(x0 - r0);
                PP_AT(DG, i, 0) = fabs(g[i] - r0);
            }
        }
        if (k < N) {
            double u[2], f[NX], sp, cp, td;
            u[0] = fmin(fmax(WS ? (double)MPC_K(P.Z, NZ, 0, 0) : src.z(bl, k, 0), PLt[k * NZ]), PHt[k * NZ]);
            {
                const double u1 = WS ? (double)MPC_K(P.Z, NZ, 0, 1) : src.z(bl, k, 1);
                u[1] = (k == 0) ? push_in(u1, A0[bl], A0[bx + bl]) : fmin(fmax(u1, PLt[k * NZ + 1]), PHt[k * NZ + 1]);      // (a_0: per-instance bounds)
            }
            PP_AT(IN, 0, k) = u[0];
            PP_AT(IN, 1, k) = u[1];
            ode_eval<NX>(P, g, u, f, sp, cp, td);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const double raw = f[i] * dt + g[i];
                const double gn = fmin(fmax(WS ? (double)MPC_K(P.Z, NZ, 1, 2 + i) : src.z(bl, k + 1, 2 + i), PLt[(k + 1) * NZ + 2 + i]), PHt[(k + 1) * NZ + 2 + i]);
                PP_AT(DG, i, k + 1) = fabs(gn - raw);
            }
        }
    }
    __syncthreads();
    PS_STAMP(2);
    // the sum of the guess's defects (its rows are complete): by the first stage-thread of the SECOND wavefront, which has nothing to do in the scans
    // -- off the critical path; the same order of additions as ever (stage-major)
    auto defect_sum = [&](const double* D, double th, const int ka, const int kb) {         // stages ka ... kb - 1 added to th
        for (int k0 = ka; k0 < kb; k0 += 4) {          // (four stages' reads in flight)
            double v[4][NX];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = (k0 + j < kb) ? k0 + j : kb - 1;
#pragma unroll
                for (int i = 0; i < NX; ++i) v[j][i] = PP_AT(D, i, k);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k0 + j < kb) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) th += v[j][i];
                }
            }
        }
        return th;
    };
    // (a third of the stages in front of each of the three scans, the running sum carried in a register: hidden behind the chains)
    const int kdg = (64 / bx <= N) ? 64 / bx : 1;
    const bool sums_dg = valid && c.k == kdg;
    double th_g = 0.0;
    int dg_part = 0;
    if (valid && c.k == 0) A0[3 * bx + bl] = 0.0;              // (set to 1 by a stage thread that finds a rolled-out state clipped)
    // one scan: lanes of stage-thread `which` (0 or 1) carry state s0 / s1 through the stages from the increments IN[which].  All that is sequential is
    // the chain x -> fma -> clip against limits fetched eight stages ahead; the clipping defects are formed afterwards, every stage thread its own
    // (the same two operations on the same operands: same bits as inside the chain).
    bool clipped = false;
    auto scan = [&](int s0, int s1, int nwhich) {
        if (sums_dg) {
            th_g = defect_sum(DG, th_g, dg_part * S / 3, (dg_part + 1) * S / 3);
            if (++dg_part == 3) A0[2 * bx + bl] = th_g;
        }
        if (valid && c.k < nwhich) {
            const int si = (c.k == 0) ? s0 : s1;
            const double* in = IN + c.k * SB;
            double x = PP_AT(XR, si, 0);
            int k0 = 0;
            for (; k0 + 8 <= N; k0 += 8) {
                double inc[8], lo[8], hi[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int k = k0 + j; inc[j] = in[k * bx + bl]; lo[j] = PLt[(k + 1) * NZ + 2 + si]; hi[j] = PHt[(k + 1) * NZ + 2 + si]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    x = fmin(fmax(__builtin_fma(inc[j], dt, x), lo[j]), hi[j]);              // (= push_in(raw, lb, ub): the limits are push_limits')
                    PP_AT(XR, si, k0 + j + 1) = x;
                }
            }
            if (k0 < N) {
                double inc[8], lo[8], hi[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int k = (k0 + j < N) ? k0 + j : N - 1; inc[j] = in[k * bx + bl]; lo[j] = PLt[(k + 1) * NZ + 2 + si]; hi[j] = PHt[(k + 1) * NZ + 2 + si]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (k0 + j < N) {
                        x = fmin(fmax(__builtin_fma(inc[j], dt, x), lo[j]), hi[j]);
                        PP_AT(XR, si, k0 + j + 1) = x;
                    }
                }
            }
        }
        __syncthreads();
        if (valid && c.k < N) {
            for (int w = 0; w < nwhich; ++w) {
                const int si = w ? s1 : s0;
                const double raw = __builtin_fma(PP_AT(IN + w * SB, 0, c.k), dt, PP_AT(XR, si, c.k));
                const double dr = fabs(PP_AT(XR, si, c.k + 1) - raw);
                PP_AT(DR, si, c.k + 1) = dr;
                clipped = clipped || dr != 0.0;
            }
        }
    };
    scan(2, 3, 2);                                            // delta, v from the controls
    PS_STAMP(3);
    if (valid && c.k < N) {
        const double dl = PP_AT(XR, 2, c.k), v = PP_AT(XR, 3, c.k);
        PP_AT(IN, 0, c.k) = v / P.wheelbase * mpc_tan(dl);            // (the functions of ode_eval: same bits as the two-chain kernel)
        PP_AT(IN, 1, c.k) = v;
    }
    __syncthreads();
    scan(4, 5, NX == 6 ? 2 : 1);                              // psi (and the progress state)
    PS_STAMP(4);
    if (valid && c.k < N) {
        double sp, cp;
        mpc_sincos(PP_AT(XR, 4, c.k), sp, cp);
        const double v = PP_AT(XR, 3, c.k);
        PP_AT(IN, 0, c.k) = v * cp;
        PP_AT(IN, 1, c.k) = v * sp;
    }
    __syncthreads();
    scan(0, 1, 2);                                            // x, y
    PS_STAMP(5);
    if (valid) {
#pragma unroll
        for (int i = 0; i < NX; ++i) MPC_K(P.ROLL, NX, 0, i) = PP_AT(XR, i, c.k);
        if (clipped) A0[3 * bx + bl] = 1.0;
    }
    __syncthreads();
    PS_STAMP(6);
    if (valid && c.k == 0) {
        // the rollout's defects: |x_0 - r_0| and what the clipping took -- zeros unless some stage thread said otherwise (adding them changes nothing)
        double th = 0.0;
        if (A0[3 * bx + bl] != 0.0) th = defect_sum(DR, 0.0, 0, S);
        else {
#pragma unroll
            for (int i = 0; i < NX; ++i) th += PP_AT(DR, i, 0);
        }
        prestart_decide<NX>(P, c.b, frow, a0lb, a0ub, A0[2 * bx + bl], th);
    }
#undef PP_AT
#undef PS_STAMP
}
template <int NX>
__global__ void __launch_bounds__(1024) k_prestart_par(const Params Pk) {
    const PRef P(Pk);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    prestart_par_block<NX>(P, (blockIdx.x + (uint32_t)P.tile0 * (64u / (uint32_t)P.bx)) * (uint32_t)P.bx, sm);
}
// ingest + start-point safeguard + start iterate of a block of bx instances in ONE launch (the three have the same thread mapping; what one
// leaves in the workspace -- rollout, per-instance bounds of a_0, the verdict -- comes back from this CU's own write-through L1 / the L2)
#ifndef MPC_KSTART_STOP
#define MPC_KSTART_STOP 0
#endif
#ifndef MPC_KSTART_OCC
#define MPC_KSTART_OCC 2
#endif
// zero_p / zero_n: the OTHER control block of the loop kernels, zeroed here for the next solve (the two blocks alternate: a solve in steady state has
// no fill of its own anywhere on its stream)
template <int NX>
__global__ void __launch_bounds__(256, MPC_KSTART_OCC) k_start(const Params Pk, const int n_mult, const int n_z, const int stash_rows, uint32_t* zero_p, const uint32_t zero_n) {
    const PRef P(Pk);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int or_slots[2][8];
    if (zero_p != nullptr && blockIdx.x == 0)
        for (uint32_t q = threadIdx.x; q < zero_n; q += blockDim.x) zero_p[q] = 0u;
    uint32_t blk = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
    const uint32_t b0 = (blk + (uint32_t)P.tile0 * (64u / (uint32_t)P.bx)) * (uint32_t)P.bx;
#define KS_STAMP(i) do { if (P.DBG && threadIdx.x == 0) P.DBG[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    KS_STAMP(11);
    {
        // ---- the caller's rows of this block (x0, the X_ref part of p: row-major [B][n_w]) -> LDS -> the tile-major Z / REF rows of the
        //      workspace: what k_ingest does for a whole batch, here for the block's own bx instances (consecutive rows: coalesced reads)
        constexpr int NZ = NX + 2;
        const int N = P.N, nw = 2 * N + NX * (N + 1), bx = P.bx, t = threadIdx.x;
        // LDS: [the safeguard's tables and scan rows (prestart_par_block) | the block's rows of x0 | of the X_ref part of p]: the safeguard reads the
        // guess and the reference from the last two, so it does not wait for the stores below
        const size_t pre_doubles = prestart_doubles(NX, N + 1, bx);
        // (the safeguard's region starts where stage_block keeps its bounds table -- behind the reduction scratch -- and begins with that table: the
        //  start iterate finds it in place)
        double* const pre = lds + (blockDim.x >> 6) * 10 * bx;
        double* rx = pre + ((pre_doubles + 1) & ~(size_t)1);     // [bx][nw]
        double* rp = rx + bx * nw;                               // [bx][nw - 2N]   (the U_ref part of p is not used by the NLP, optimizer.py:507-511)
        const int nrow = ((int)b0 + bx <= P.B) ? bx : (P.B > (int)b0 ? P.B - (int)b0 : 0);
        const int npx = nw - 2 * N;
        // (every load of a thread requested before its first LDS write: batched_fill; the rows of p one per trip element -- no division)
        batched_fill<8>(nrow * nw, t, (int)blockDim.x, [&](int q) { return MPC_GP(P.x0, (size_t)b0 * nw + q); }, [&](int q, double v) { rx[q] = v; });
        for (int r0 = 0; r0 < nrow; r0 += 8) {
            for (int c0 = 0; c0 < npx; c0 += (int)blockDim.x) {
                const int col = c0 + t;
                double v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = (r0 + r < nrow && col < npx) ? (double)MPC_GP(P.p, ((size_t)b0 + r0 + r) * nw + 2 * N + col) : 0.0;
#pragma unroll
                for (int r = 0; r < 8; ++r) if (r0 + r < nrow && col < npx) rp[(r0 + r) * npx + col] = v[r];
            }
        }
        __syncthreads();
        KS_STAMP(12);
#if MPC_KSTART_STOP == 3
        return;
#endif
        struct { int b, k; } c;
        c.k = t / bx;
        const int bl = t & (bx - 1);
        c.b = (int)b0 + bl;
        if (c.k <= N && c.b < P.B) {
            double z[NZ], r[NX];
            const double* xr = rx + bl * nw;
            z[0] = (c.k < N) ? xr[2 * c.k] : 0.0;
            z[1] = (c.k < N) ? xr[2 * c.k + 1] : 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) { z[2 + i] = xr[2 * N + NX * c.k + i]; r[i] = rp[bl * npx + NX * c.k + i]; }
            ws_store_rows<NZ>(MPC_ROWS(MPC_K(P.Z, NZ, 0, e)), z);
            ws_store_rows<NX>(MPC_ROWS(MPC_K(P.REF, NX, 0, e)), r);
        }
        struct FromLds {
            const double *rx, *rp;
            int nw, npx, N;
            __device__ __forceinline__ double z(int bl, int k, int i) const { return i < 2 ? (k < N ? rx[bl * nw + 2 * k + i] : 0.0) : rx[bl * nw + 2 * N + NX * k + (i - 2)]; }
            __device__ __forceinline__ double ref0(int bl, int i) const { return rp[bl * npx + i]; }
        } src{rx, rp, nw, npx, N};
        KS_STAMP(13);
#if MPC_KSTART_STOP == 4
        return;
#endif
        prestart_par_block<NX, FromLds>(P, b0, pre, src);
    }
    KS_STAMP(14);
#if MPC_KSTART_STOP == 1
    return;
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KS_STAMP(15);
#undef KS_STAMP
#if MPC_KSTART_STOP == 2
    return;
#endif
    stage_block<NX, true, 256>(P, n_mult, n_z, stash_rows, b0, ~0ull, lds, or_slots, true, nullptr, true);
}


// ---------------------------------------------------------------------------------------------------------------
// k_ingest / k_egest: LDS-tiled transposes between the caller's row-major [B][n_w] buffers (optimizer.py:550 order)
// and the tile-major workspace.  One workgroup per tile of 64 instances; both the global reads and the global
// writes are 512-byte contiguous per wavefront.
//   ingest:  Z[(k, i)] <- x0 (raw),  REF[(k, i)] <- X_ref part of p        egest:  x_out <- Z, plus status/iters/kkt
// ---------------------------------------------------------------------------------------------------------------
template <int NX>
__device__ __forceinline__ uint32_t zrow_of_col(int col, int N) {          // decision-vector column -> row of Z
    constexpr int NZ = NX + 2;
    if (col < 2 * N) return (uint32_t)(col >> 1) * MPC_EV(NZ) + (uint32_t)(col & 1);
    const int cx = col - 2 * N;
    return (uint32_t)(cx / NX) * MPC_EV(NZ) + 2u + (uint32_t)(cx % NX);
}

// grid = (tiles, 64-column chunks): every workgroup moves one 64 x 64 block, so a batch of 64 tiles is ~450 workgroups
template <int NX>
__global__ void __launch_bounds__(256) k_ingest(const Params Pk) {
    const PRef P(Pk);
    __shared__ double tile[64][65];
    const int N = P.N, nw = 2 * N + NX * (N + 1);
    const uint32_t tl = blockIdx.x + (uint32_t)P.tile0, t0 = tl * 64u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double* Zt = P.Z + (size_t)tl * P.tile_elems;
    double* Rt = P.REF + (size_t)tl * P.tile_elems;
    const int nc0 = (nw + 63) / 64;                          // chunks of x0; the rest are chunks of the X_ref part of p
    const int pass = (int)blockIdx.y < nc0 ? 0 : 1;
    const int c0 = pass == 0 ? (int)blockIdx.y * 64 : 2 * N + ((int)blockIdx.y - nc0) * 64;
    const double* src = pass == 0 ? P.x0 : P.p;
    for (int r = w; r < 64; r += 4) {
        const int col = c0 + lane;
        const uint32_t b = t0 + (uint32_t)r;
        tile[r][lane] = (col < nw && b < (uint32_t)P.B) ? src[(size_t)b * nw + col] : 0.0;
    }
    __syncthreads();
    for (int cc = w; cc < 64; cc += 4) {
        const int col = c0 + cc;
        if (col < nw) {
            if (pass == 0) Zt[mpc_prow(zrow_of_col<NX>(col, N)) + 2u * lane] = tile[lane][cc];
            else Rt[mpc_prow((uint32_t)((col - 2 * N) / NX) * MPC_EV(NX) + (uint32_t)((col - 2 * N) % NX)) + 2u * lane] = tile[lane][cc];
        }
    }
    // u rows of the terminal stage do not exist in x0
    constexpr int NZ = NX + 2;
    if (blockIdx.y == 0 && threadIdx.x < 128) Zt[mpc_prow((uint32_t)N * MPC_EV(NZ) + (threadIdx.x >> 6)) + 2u * lane] = 0.0;
}

template <int NX>
// zero_p / zero_n: the pipeline's OTHER control block, zeroed here for the next solve (the two blocks alternate: a solve in steady state has
// no fill of its own in front of its persistent launch)
__global__ void __launch_bounds__(256) k_egest(const Params Pk, const uint32_t* skip_if, uint32_t* fail_count, uint32_t* zero_p, const uint32_t zero_n) {
    const PRef P(Pk);
    __shared__ double tile[64][65];
    if (zero_p != nullptr && blockIdx.x == 0 && blockIdx.y == 0)
        for (uint32_t q = threadIdx.x; q < zero_n; q += 256u) zero_p[q] = 0u;
    if (skip_if != nullptr && *skip_if != 0u) return;        // abandoned pipeline launch: the host starts over, the caller's buffers stay untouched
    const int N = P.N, nw = 2 * N + NX * (N + 1);
    const uint32_t tl = blockIdx.x + (uint32_t)P.tile0, t0 = tl * 64u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double* Zt = P.Z + (size_t)tl * P.tile_elems;
    const int c0 = (int)blockIdx.y * 64;
    for (int cc = w; cc < 64; cc += 4) {
        const int col = c0 + cc;
        tile[lane][cc] = (col < nw) ? Zt[mpc_prow(zrow_of_col<NX>(col, N)) + 2u * lane] : 0.0;
    }
    __syncthreads();
    for (int r = w; r < 64; r += 4) {
        const int col = c0 + lane;
        const uint32_t b = t0 + (uint32_t)r;
        if (col < nw && b < (uint32_t)P.B) P.x_out[(size_t)b * nw + col] = tile[r][lane];
    }
    if (blockIdx.y != 0) return;
    if (threadIdx.x < 64) {
        const uint32_t b = t0 + (uint32_t)lane;
        int st = 1;
        if (b < (uint32_t)P.B) {
            st = P.ISC[tl * P.itile_elems + mpc_prow((uint32_t)IS_STATUS) + 2u * lane];
            if (st == ST_RUNNING) st = 0;          // iteration budget of the launch loop exhausted
            if (P.status_out) P.status_out[b] = st;
            if (P.iters_out) P.iters_out[b] = P.ISC[tl * P.itile_elems + mpc_prow((uint32_t)IS_ITERS) + 2u * lane];
            if (P.kkt_out) P.kkt_out[b] = P.SC[(size_t)tl * P.tile_elems + mpc_prow((uint32_t)SC_E0) + 2u * lane];
        }
        // how many instances of the batch ran out of iterations or stalled (the host decides from this whether a second chance is due)
        const int nbad = __popcll(__ballot((st == 0 || st == -7) ? 1 : 0));      // (NaN inputs stay NaN: not counted)
        if (fail_count != nullptr && lane == 0 && nbad) atomicAdd(fail_count, (uint32_t)nbad);
    }
}

__global__ void k_count_running(const int32_t* iws, uint32_t itile_elems, int b0, int B, int32_t* counter) {
    const int b = b0 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int run = (b < B && iws[((uint32_t)b >> 6) * itile_elems + mpc_prow((uint32_t)IS_STATUS) + 2u * ((uint32_t)b & 63u)] == ST_RUNNING) ? 1 : 0;
    const unsigned long long m = __ballot(run);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (int)__popcll(m));
}

__global__ void k_transpose_obst(const double* obst /*[B][6]*/, double* OBST /*rows of tile 0*/, int B, uint32_t tile_elems) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
#pragma unroll
    for (int i = 0; i < 6; ++i) OBST[((uint32_t)b >> 6) * tile_elems + mpc_prow((uint32_t)i) + 2u * ((uint32_t)b & 63u)] = obst[(size_t)b * 6 + i];
}

// debug/trace helper: gather 8 per-instance scalar rows into a contiguous [8][B] buffer
__global__ void k_gather_trace(const double* SC /*rows of tile 0*/, uint32_t tile_elems, int B, double* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int rows[8] = {SC_MU, SC_THETA, SC_PHI, SC_ALPHA, SC_ADU, SC_DELTA, SC_E0, SC_NTRIAL};
#pragma unroll
    for (int q = 0; q < 8; ++q) out[(size_t)q * B + b] = SC[((uint32_t)b >> 6) * tile_elems + mpc_prow((uint32_t)rows[q]) + 2u * ((uint32_t)b & 63u)];
}

// correctly rounded square root: the library sqrt is within 1 ulp; one residual correction r += (x - r*r) / (2r) with a
// fused residual makes it match the IEEE result numpy produces
__device__ __forceinline__ double sqrt_cr(double x) {
    const double r = sqrt(x);
    if (!(r > 0.0) || !isfinite(r)) return r;
    return fma(fma(-r, r, x), 0.5 / r, r);
}

// post-hoc trajectory metrics (row f4; mpc_planner.py:184-199, 279-292, configuration.py:26-37): one workgroup per ego.
// Every operation is a separately rounded IEEE operation (fp contraction off: no fused multiply-add), in the
// reference's order, so the results are bit-identical to the numpy expressions.
__global__ void __launch_bounds__(256) k_metrics(int L, int Lo, const double* traj, const double* ref_path, const double* origin_path,
                                                 const double* obst, double ego_offset, double r_sum, int all_pairs, double* deviation,
                                                 double* rmsd, double* clearance) {
#pragma clang fp contract(off)         // plain operators below, each rounded separately (HIP's rn-intrinsics are plain operators that get fused after inlining)
    __shared__ double red[256];
    const int b = blockIdx.x, t = threadIdx.x;
    const double* x = traj + (size_t)b * L * 5;
    if (deviation != nullptr && origin_path != nullptr) {
        const double* op = origin_path + (size_t)b * Lo * 2;
        for (int i = t; i < L; i += 256) {
            const double px = x[i * 5], py = x[i * 5 + 1];
            double best = INFINITY;
            int arg = 0;
            for (int q = 0; q < Lo; ++q) {                       // np.argmin: first index of the minimum
                const double dx = op[2 * q] - px, dy = op[2 * q + 1] - py;
                const double sq = dx * dx + dy * dy;
                if (sq < best) { best = sq; arg = q; }
            }
            const double ex = op[2 * arg] - px, ey = op[2 * arg + 1] - py;
            deviation[(size_t)b * L + i] = sqrt_cr(ex * ex + ey * ey);
        }
    }
    if (rmsd != nullptr && ref_path != nullptr && t == 0) {       // sequential sums in the reference's order
        const double* rp = ref_path + (size_t)b * L * 2;
        double sx = 0.0, sy = 0.0;
        for (int i = 0; i < L; ++i) {
            const double dx = rp[2 * i] - x[i * 5], dy = rp[2 * i + 1] - x[i * 5 + 1];
            sx = sx + dx * dx;
            sy = sy + dy * dy;
        }
        rmsd[(size_t)b * 2] = sqrt_cr(sx / (double)(L - 1));
        rmsd[(size_t)b * 2 + 1] = sqrt_cr(sy / (double)(L - 1));
    }
    if (clearance != nullptr) {
        double best = INFINITY;
        for (int i = t; i < L; i += 256) {
            const double sx = x[i * 5], sy = x[i * 5 + 1], psi = x[i * 5 + 4];
            double sn, cs;
            sincos(psi, &sn, &cs);
            for (int e = 0; e < 3; ++e) {
                const double sg = (e == 0) ? 0.0 : (e == 1 ? 1.0 : -1.0);
                const double ex = sx + sg * ego_offset * cs, ey = sy + sg * ego_offset * sn;
                for (int j = 0; j < 3; ++j) {
                    if (!all_pairs && j != e) continue;           // the NLP constrains circle e against circle e only (optimizer.py:395-403)
                    const double dx = ex - obst[2 * j], dy = ey - obst[2 * j + 1];
                    best = fmin(best, sqrt(dx * dx + dy * dy) - r_sum);
                }
            }
        }
        red[t] = best;
        __syncthreads();
        for (int m = 128; m > 0; m >>= 1) { if (t < m) red[t] = fmin(red[t], red[t + m]); __syncthreads(); }
        if (t == 0) clearance[b] = red[0];
    }
}

// collision / road verdict of planned trajectories (row f4; what test/test_mpc_planner.py:37-47 asks of commonroad_dc's collision
// checker): the ego rectangle (mpc_planner.py:99: 4.3 x 1.8 m, centred on the planned position, heading psi) of every step
// against the obstacle rectangles of the same time step (separating-axis test of two oriented rectangles) and against the
// drivable corridor (every corner right of the left boundary polyline and left of the right one, judged at the nearest
// segment).  One workgroup per trajectory, threads over the steps; first offending step by atomicMin.
__device__ __forceinline__ double seg_side(const double* poly, int n, double px, double py) {
    // signed side of (px, py) w.r.t. the nearest segment of the polyline: > 0 left of it, < 0 right of it
    double best = INFINITY, side = 0.0;
    for (int q = 0; q + 1 < n; ++q) {
        const double ax = poly[2 * q], ay = poly[2 * q + 1], bx = poly[2 * q + 2], by = poly[2 * q + 3];
        const double ex = bx - ax, ey = by - ay, l2 = ex * ex + ey * ey;
        double t = l2 > 0.0 ? ((px - ax) * ex + (py - ay) * ey) / l2 : 0.0;
        t = fmin(1.0, fmax(0.0, t));
        const double dx = px - (ax + t * ex), dy = py - (ay + t * ey), d2 = dx * dx + dy * dy;
        if (d2 < best) { best = d2; side = ex * (py - ay) - ey * (px - ax); }
    }
    return side;
}
__global__ void __launch_bounds__(128) k_validity(int L, const double* traj, double ego_l, double ego_w, int n_obst, const double* obst, int n_left,
                                                  const double* left, int n_right, const double* right, int32_t* first_collision,
                                                  int32_t* first_off_road) {
#pragma clang fp contract(off)
    __shared__ int s_col, s_off;
    const int b = blockIdx.x, t = threadIdx.x;
    if (t == 0) { s_col = 0x7fffffff; s_off = 0x7fffffff; }
    __syncthreads();
    const double* x = traj + (size_t)b * L * 5;
    for (int i = t; i < L; i += 128) {
        const double cx = x[i * 5], cy = x[i * 5 + 1], psi = x[i * 5 + 4];
        double sn, cs;
        sincos(psi, &sn, &cs);
        const double hl = 0.5 * ego_l, hw = 0.5 * ego_w;
        bool col = false, off = false;
        for (int o = 0; o < n_obst && !col; ++o) {
            const double* r = obst + ((size_t)o * L + i) * 5;
            if (!(r[2] > 0.0) || !(r[3] > 0.0)) continue;                  // absent at this time step
            double so, co;
            sincos(r[4], &so, &co);
            const double ol = 0.5 * r[2], ow = 0.5 * r[3], dx = r[0] - cx, dy = r[1] - cy;
            // axes: ego (cs, sn), (-sn, cs); obstacle (co, so), (-so, co)
            const double c00 = fabs(cs * co + sn * so), c01 = fabs(-cs * so + sn * co);       // |ego axis . obstacle axis|
            bool sep = fabs(dx * cs + dy * sn) > hl + ol * c00 + ow * c01;
            sep = sep || fabs(-dx * sn + dy * cs) > hw + ol * c01 + ow * c00;
            sep = sep || fabs(dx * co + dy * so) > ol + hl * c00 + hw * c01;
            sep = sep || fabs(-dx * so + dy * co) > ow + hl * c01 + hw * c00;
            col = !sep;
        }
        if (n_left > 1 || n_right > 1) {
            for (int q = 0; q < 4 && !off; ++q) {
                const double sl = (q & 1) ? -hl : hl, sw = (q & 2) ? -hw : hw;
                const double px = cx + sl * cs - sw * sn, py = cy + sl * sn + sw * cs;
                if (n_left > 1 && seg_side(left, n_left, px, py) > 0.0) off = true;
                if (n_right > 1 && seg_side(right, n_right, px, py) < 0.0) off = true;
            }
        }
        if (col) atomicMin(&s_col, i);
        if (off) atomicMin(&s_off, i);
    }
    __syncthreads();
    if (t == 0) {
        first_collision[b] = s_col == 0x7fffffff ? -1 : s_col;
        first_off_road[b] = s_off == 0x7fffffff ? -1 : s_off;
    }
}

// FORCES-mode stage functions (row a11; FORCESNLPsolver_interface.c:41-198 / FORCESNLPsolver_model.c:75-1756, the model of
// optimizer.py:91-245): per instance  z = [deltaDot, aLong, x, y, delta, v, psi],  p = [x_ref, y_ref, v_des, psi_ref, 3 obstacle
// circle centres]  ->  stage cost f and gradient (7), one RK4 step c (5) with Jacobian (5x7, forward sensitivities through
// the four stages), inequality functions h (10: friction circle, 9 squared circle distances) with Jacobian (10x7).
// One thread per instance; row-major outputs.
struct ForcesArgs {
    int32_t B, terminal;
    double dt, l, wb, rho, Q[5], R[2], Pt[5];
    const double *z, *p;
    double *f, *grad_f, *c, *jac_c, *h, *jac_h;
};
__device__ __forceinline__ void forces_ode(const double* x, const double* u, double l, double* f, double* F /*[4]: (0,3),(0,4),(1,3),(1,4)*/, double& F42, double& F43) {
    double sn, cs;
    sincos(x[4], &sn, &cs);
    const double td = tan(x[2]);
    f[0] = x[3] * cs; f[1] = x[3] * sn; f[2] = u[0]; f[3] = u[1]; f[4] = x[3] / l * td;
    F[0] = cs; F[1] = -x[3] * sn; F[2] = sn; F[3] = x[3] * cs;
    F42 = x[3] / l * (1.0 + td * td);
    F43 = td / l;
}
__global__ void __launch_bounds__(128) k_forces_stage(const ForcesArgs A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.B) return;
    double z[7], p[10];
#pragma unroll
    for (int i = 0; i < 7; ++i) z[i] = A.z[(size_t)b * 7 + i];
#pragma unroll
    for (int i = 0; i < 10; ++i) p[i] = A.p[(size_t)b * 10 + i];
    // ---- cost (optimizer.py:158-194)
    const double* w = A.terminal ? A.Pt : A.Q;
    const double r[5] = {z[2] - p[0], z[3] - p[1], z[4], z[5] - p[2], z[6] - p[3]};
    double f = 0.0, gf[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 5; ++i) { f += w[i] * r[i] * r[i]; gf[2 + i] = 2.0 * w[i] * r[i]; }
    if (!A.terminal) {
        f += A.R[0] * z[0] * z[0] + A.R[1] * z[1] * z[1];
        gf[0] = 2.0 * A.R[0] * z[0];
        gf[1] = 2.0 * A.R[1] * z[1];
    }
    if (A.f) A.f[b] = f;
    if (A.grad_f) {
#pragma unroll
        for (int i = 0; i < 7; ++i) A.grad_f[(size_t)b * 7 + i] = gf[i];
    }
    // ---- RK4 step and its Jacobian (optimizer.py:91-98): tangents T = d(.)/dz, 5 x 7
    if (!A.terminal && (A.c || A.jac_c)) {
        const double* u = z;
        const double* x = z + 2;
        double xs[5], k[5], acc[5], dk[5][7], dacc[5][7], dxs[5][7];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            xs[i] = x[i];
            acc[i] = 0.0;
#pragma unroll
            for (int j = 0; j < 7; ++j) { dxs[i][j] = (j == i + 2) ? 1.0 : 0.0; dacc[i][j] = 0.0; dk[i][j] = 0.0; }
        }
        const double aw[4] = {0.0, 0.5, 0.5, 1.0}, bw[4] = {1.0, 2.0, 2.0, 1.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s > 0) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    xs[i] = x[i] + aw[s] * A.dt * k[i];
#pragma unroll
                    for (int j = 0; j < 7; ++j) dxs[i][j] = ((j == i + 2) ? 1.0 : 0.0) + aw[s] * A.dt * dk[i][j];
                }
            }
            double F[4], F42, F43;
            forces_ode(xs, u, A.l, k, F, F42, F43);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                dk[0][j] = F[0] * dxs[3][j] + F[1] * dxs[4][j];
                dk[1][j] = F[2] * dxs[3][j] + F[3] * dxs[4][j];
                dk[2][j] = (j == 0) ? 1.0 : 0.0;
                dk[3][j] = (j == 1) ? 1.0 : 0.0;
                dk[4][j] = F42 * dxs[2][j] + F43 * dxs[3][j];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                acc[i] += bw[s] * k[i];
#pragma unroll
                for (int j = 0; j < 7; ++j) dacc[i][j] += bw[s] * dk[i][j];
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (A.c) A.c[(size_t)b * 5 + i] = x[i] + A.dt / 6.0 * acc[i];
            if (A.jac_c) {
#pragma unroll
                for (int j = 0; j < 7; ++j) A.jac_c[((size_t)b * 5 + i) * 7 + j] = ((j == i + 2) ? 1.0 : 0.0) + A.dt / 6.0 * dacc[i][j];
            }
        }
    }
    // ---- inequalities (optimizer.py:121-149)
    if (A.h || A.jac_h) {
        double h[10], jh[10][7];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
#pragma unroll
            for (int j = 0; j < 7; ++j) jh[i][j] = 0.0;
        }
        const double td = tan(z[4]);
        const double q = z[5] * z[5] * td / A.wb;                 // v * psi_dot
        h[0] = z[1] * z[1] + q * q;
        jh[0][1] = 2.0 * z[1];
        jh[0][4] = 2.0 * q * z[5] * z[5] * (1.0 + td * td) / A.wb;
        jh[0][5] = 2.0 * q * 2.0 * z[5] * td / A.wb;
        double sn, cs;
        sincos(z[6], &sn, &cs);
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const double sg = (e == 0) ? 0.0 : (e == 1 ? 1.0 : -1.0);
            const double ex = z[2] + sg * A.rho * cs, ey = z[3] + sg * A.rho * sn;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double dx = ex - p[4 + 2 * j], dy = ey - p[5 + 2 * j];
                const int row = 1 + 3 * e + j;
                h[row] = dx * dx + dy * dy;
                jh[row][2] = 2.0 * dx;
                jh[row][3] = 2.0 * dy;
                jh[row][6] = 2.0 * dx * (-sg * A.rho * sn) + 2.0 * dy * (sg * A.rho * cs);
            }
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            if (A.h) A.h[(size_t)b * 10 + i] = h[i];
            if (A.jac_h) {
#pragma unroll
                for (int j = 0; j < 7; ++j) A.jac_h[((size_t)b * 10 + i) * 7 + j] = jh[i][j];
            }
        }
    }
}

// FORCES-mode SQP step (row f3; phases in mpc_forces_qp.h): one thread per (instance, stage), a workgroup holds all N stages of
// IB instances (thread t: stage t / IB, instance t % IB -- the threads of one stage read consecutive doubles of a workspace
// row).  Stage-local work runs for all stages at once; the Riccati recursion hands the turn from stage to stage between
// workgroup barriers, cost-to-go and step travelling through the workspace rows of the neighbouring stage (same workgroup,
// same CU: ordered by the barrier).  Per-instance reductions combine the stage partials in stage order.
template <class Combine>
__device__ __forceinline__ FqRed fq_block_reduce(const FqRed& part, FqRed acc, Combine comb, double* lds, int IB, int N) {
    const int t = threadIdx.x, T = blockDim.x, i = t % IB;
    lds[t] = part.a; lds[T + t] = part.b; lds[2 * T + t] = part.c; lds[3 * T + t] = part.d;
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        const FqRed p{lds[k * IB + i], lds[T + k * IB + i], lds[2 * T + k * IB + i], lds[3 * T + k * IB + i]};
        comb(acc, p);
    }
    __syncthreads();
    return acc;
}
__global__ void __launch_bounds__(256) k_forces_qp(const ForcesQpArgs A, const int IB) {
    extern __shared__ __attribute__((aligned(16))) double fq_lds[];
    const int t = threadIdx.x, N = A.N;
    FqCtx c{};
    c.k = t / IB;
    c.b = (int)blockIdx.x * IB + t % IB;
    c.valid = (c.k < N) && (c.b < A.B);
    c.run = c.valid;
    c.it = 0; c.status = 0; c.kkt = INFINITY; c.gscale = 1.0;
    c.t = t; c.T = (int)blockDim.x; c.IB = IB;
    c.loc = fq_lds + 4 * blockDim.x;                       // behind the reduction scratch
#if defined(__HIP_DEVICE_COMPILE__)
    c.rs = mpc_rsrc(A.ws, A.ws_bytes);
#endif
    FqRed part, tot;
    fq_build(A, c, part);
    tot = fq_block_reduce(part, FqRed{1.0, 0.0, 0.0, 0.0}, [](FqRed& a, const FqRed& p) { fq_max_combine(a, p); }, fq_lds, IB, N);
    c.gscale = tot.a;
    for (;;) {
        fq_residual(A, c, part);
        tot = fq_block_reduce(part, FqRed{0.0, 0.0, 0.0, 0.0}, [](FqRed& a, const FqRed& p) { fq_residual_combine(a, p); }, fq_lds, IB, N);
        const double n_rows = tot.b;
        const double mu = fq_decide(A, c, tot);
        if (!__syncthreads_or(c.run ? 1 : 0)) break;
        double sigma_mu = 0.0;
        for (int corr = 0; corr < 2; ++corr) {
            if (corr) {
                fq_steplen(A, c, 1.0, part);
                tot = fq_block_reduce(part, FqRed{1.0, 1.0, 0.0, 0.0}, [](FqRed& a, const FqRed& p) { fq_steplen_combine(a, p); }, fq_lds, IB, N);
                const double a_aff = fmin(tot.a, tot.b);
                fq_mu_aff(A, c, a_aff, part);
                tot = fq_block_reduce(part, FqRed{0.0, 0.0, 0.0, 0.0}, [](FqRed& a, const FqRed& p) { fq_sum_combine(a, p); }, fq_lds, IB, N);
                const double mu_aff = tot.a / n_rows, sg = mu_aff / mu;
                sigma_mu = sg * sg * sg * mu;
            }
            fq_newton_prep(A, c, corr != 0, sigma_mu);
            // (the turns hand P_k, p_k, dx_k on through LDS and read nothing but LDS and registers: their barriers order LDS traffic only --
            //  __syncthreads would also wait for every global store of the turn to be acknowledged, forty times per iteration)
            __syncthreads();
            for (int step = 0; step < N; ++step) {
                if (c.k == N - 1 - step) fq_newton_back(A, c, corr != 0);
                lds_barrier();
            }
            for (int step = 0; step < N; ++step) {
                if (c.k == step) fq_newton_fwd(A, c);
                lds_barrier();
            }
            __syncthreads();                 // (the step rows the forward turns stored are read by the row phase below)
            fq_newton_rows(A, c, corr != 0, sigma_mu);
        }
        fq_steplen(A, c, 0.995, part);
        tot = fq_block_reduce(part, FqRed{1.0, 1.0, 0.0, 0.0}, [](FqRed& a, const FqRed& p) { fq_steplen_combine(a, p); }, fq_lds, IB, N);
        fq_update(A, c, tot.a, tot.b);
        __syncthreads();
    }
    fq_output(A, c);
}

// closed-loop driver around the solve (row f1): one instance per thread, row-major buffers
__global__ void k_loop_setup(const LoopArgs A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < A.B) loop_setup_instance(A, b);
}
// after solve i: what loop_advance_instance (mpc_closed_loop.h, the form the CPU harness steps through) does for one
// instance, laid out for coalesced rows -- one workgroup per instance, the threads run over the n_w columns of the three
// row-major rows (solution in, warm start and parameter vector out); thread 0 records the step and integrates the plant.
// Same values bit for bit (the columns are copies, the plant step and the noise samples are the same code).
__global__ void __launch_bounds__(128) k_loop_advance(const Params Pk, const LoopArgs A, const int i) {
    const PRef P(Pk);
    __shared__ double cur_s[6];
    if (A.abort_flag != nullptr && *A.abort_flag != 0u) return;       // a solve of this loop was abandoned: the host replays the loop
    const int b = (int)blockIdx.x, t = (int)threadIdx.x;
    const int N = A.N, nx = A.nx, nw = 2 * N + nx * (N + 1);
    const double* xo = A.x_out + (size_t)b * nw;
    if (t == 0) {
        double cur[6], u[2], f[6], s, c, td;
        for (int q = 0; q < nx; ++q) cur[q] = A.state[(size_t)b * nx + q];
        u[0] = xo[0] + loop_noise(A, b, i, 0, 0);
        u[1] = xo[1] + loop_noise(A, b, i, 1, 0);
        for (int q = 0; q < 5; ++q) A.traj[((size_t)b * A.L + i) * 5 + q] = cur[q];
        A.ctrl[((size_t)b * A.L + i) * 2 + 0] = u[0];
        A.ctrl[((size_t)b * A.L + i) * 2 + 1] = u[1];
        if (A.step_status) A.step_status[(size_t)b * A.L + i] = A.status ? A.status[b] : 0;
        if (nx == 5) ode_eval<5>(P, cur, u, f, s, c, td); else ode_eval<6>(P, cur, u, f, s, c, td);
        for (int q = 0; q < nx; ++q) { cur[q] = cur[q] + P.dt * f[q]; A.state[(size_t)b * nx + q] = cur[q]; cur_s[q] = cur[q]; }
    }
    __syncthreads();
    double* x0 = A.x0 + (size_t)b * nw;
    double* p = A.p + (size_t)b * nw;
    const double vd = A.vdes[b];
    const int shift = (i >= A.L - N) ? (i - (A.L - N) + 1) : 0;                 // frozen tail of the reference window (loop_write_reference)
    for (int q = t; q < nw; q += (int)blockDim.x) {
        double v0, vp = 0.0;
        if (q < N) {
            const int src = (q + 1 < N) ? q + 1 : N - 1;
            v0 = xo[2 * src] + (A.noise_mode == 1 ? loop_noise(A, b, i, 0, src) : 0.0);         // all steering rates ...
        } else if (q < 2 * N) {
            const int src = (q - N + 1 < N) ? q - N + 1 : N - 1;
            v0 = xo[2 * src + 1] + (A.noise_mode == 1 ? loop_noise(A, b, i, 1, src) : 0.0);     // ... then all accelerations
        } else {
            const int r = q - 2 * N, k = r / nx, c = r - nx * k;
            v0 = xo[2 * N + nx * ((k + 1 <= N) ? k + 1 : N) + c];
            if (k == 0) vp = cur_s[c];
            else {
                const int idx = i + k - shift;                                    // = i_done + (k - 1) + 1 - shift
                vp = (c == 0) ? A.path[((size_t)b * A.Lp + idx) * 2] : (c == 1) ? A.path[((size_t)b * A.Lp + idx) * 2 + 1]
                   : (c == 3) ? vd : (c == 4) ? A.orient[(size_t)b * A.Lp + idx] : 0.0;
            }
        }
        x0[q] = v0;
        p[q] = vp;
    }
}
// sticky abort word of an asynchronous closed loop: set once any of its pipeline launches raised its own abort word
__global__ void k_loop_sticky(const uint32_t* pipe_abort, uint32_t* loop_abort) {
    if (*pipe_abort != 0u) *loop_abort = 1u;
}

// FORCES-mode closed loop (mpc_closed_loop.h: ForcesLoopArgs), one instance per thread
__global__ void k_floop_setup(const ForcesLoopArgs A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < A.B) forces_loop_setup_instance(A, b);
}
__global__ void k_floop_params(const ForcesLoopArgs A, const int k) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < A.B) forces_loop_params_instance(A, b, k);
}
__global__ void k_floop_advance(const ForcesLoopArgs A, const int k) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < A.B) forces_loop_advance_instance(A, b, k);
}

// ---- second chance for stalled instances (see rescue_dev on the host side) ---------------------------------------------
// ordered list of the instances whose status is not "converged": one workgroup, ballot-based compaction
__global__ void __launch_bounds__(1024) k_rescue_select(const int32_t* status, int B, int32_t* idx, int32_t* count) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 1024) {
        const int b = b0 + t;
        const int bad = (b < B && (status[b] == 0 || status[b] == -7)) ? 1 : 0;          // iteration limit / no progress; a NaN (-6) would stay one
        const unsigned long long m = __ballot(bad);
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int q = 0; q < w; ++q) off += wsum[q];
        if (bad) idx[off + __popcll(m & ((1ull << lane) - 1ull))] = b;
        __syncthreads();
        if (t == 0) { int s = 0; for (int q = 0; q < 16; ++q) s += wsum[q]; base += s; }
        __syncthreads();
    }
    if (t == 0) *count = base;
}
// rows of the selected instances -> compact sub-batch (first call), converged rows of a level -> warm start of the next
__global__ void k_rescue_gather(const int32_t* idx, int nw, const double* x0, const double* p, const double* obst, double* xs, double* ps, double* os,
                                int32_t* it_acc) {
    const int j = blockIdx.x, b = idx[j];
    for (int q = threadIdx.x; q < nw; q += blockDim.x) { xs[(size_t)j * nw + q] = x0[(size_t)b * nw + q]; ps[(size_t)j * nw + q] = p[(size_t)b * nw + q]; }
    if (obst != nullptr && threadIdx.x < 6) os[(size_t)j * 6 + threadIdx.x] = obst[(size_t)b * 6 + threadIdx.x];
    if (threadIdx.x == 0) it_acc[j] = 0;
}
__global__ void k_rescue_carry(int nw, const int32_t* st, const int32_t* it, const double* out, double* xs, int32_t* it_acc) {
    const int j = blockIdx.x;
    if (threadIdx.x == 0) it_acc[j] += it[j];
    if (st[j] != 1) return;
    for (int q = threadIdx.x; q < nw; q += blockDim.x) xs[(size_t)j * nw + q] = out[(size_t)j * nw + q];
}
// result of the last level (the ORIGINAL problem) back into the caller's rows -- only where it converged
__global__ void k_rescue_scatter(const int32_t* idx, int nw, const int32_t* st, const double* out, const double* kkt, const int32_t* it_acc, double* x_out,
                                 int32_t* status, int32_t* iters, double* kkt_out) {
    const int j = blockIdx.x, b = idx[j];
    if (st[j] != 1) return;
    for (int q = threadIdx.x; q < nw; q += blockDim.x) x_out[(size_t)b * nw + q] = out[(size_t)j * nw + q];
    if (threadIdx.x == 0) {
        status[b] = 1;
        if (iters) iters[b] += it_acc[j];
        if (kkt_out) kkt_out[b] = kkt[j];
    }
}

template <int NX>
__global__ void k_plant_step(const Params Pk, const double* x, const double* u, double* xn, int B, int integrator) {
    const PRef P(Pk);
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double xs[NX], us[2], f[NX], s, c, td;
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = x[(size_t)b * NX + i];
    us[0] = u[(size_t)b * 2];
    us[1] = u[(size_t)b * 2 + 1];
    const double h = P.dt;
    if (integrator == 0) {
        ode_eval<NX>(P, xs, us, f, s, c, td);
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[(size_t)b * NX + i] = xs[i] + h * f[i];
    } else {
        double k1[NX], k2[NX], k3[NX], k4[NX], tmp[NX];
        ode_eval<NX>(P, xs, us, k1, s, c, td);
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = xs[i] + 0.5 * h * k1[i];
        ode_eval<NX>(P, tmp, us, k2, s, c, td);
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = xs[i] + 0.5 * h * k2[i];
        ode_eval<NX>(P, tmp, us, k3, s, c, td);
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = xs[i] + h * k3[i];
        ode_eval<NX>(P, tmp, us, k4, s, c, td);
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[(size_t)b * NX + i] = xs[i] + h / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    }
}

}  // namespace

// ============================================================================================== host
struct mpc_handle {
    HostProblem hp;
    std::string err;
    int device = 0;
    // device workspace
    size_t cap_Bp = 0;
    bool ws_mailbox = false;            // the workspace carries the instance-major mailbox section (wants_mailbox at allocation time)
    double* d_ws = nullptr;
    int32_t* d_iws = nullptr;
    double *d_LB = nullptr, *d_UB = nullptr;
    static constexpr int MAX_POLL_IT = 1024;
    unsigned long long* d_tile_mask = nullptr;
    double* d_state = nullptr;         // [B,5] plant state of the closed-loop driver
    size_t cap_state = 0;
    size_t tile_mask_cap = 0;
    uint32_t* d_pipe = nullptr;        // control blocks of the single-launch pipeline (k_pipeline): two halves of pipe_words words that alternate --
                                       // the output kernel of a solve zeroes the other half for the next one
    int pipe_flip = 0;
    size_t pipe_clean[2] = {0, 0};     // leading words of each half known to be zero
    size_t pipe_words = 0;
    uint32_t* h_pipe = nullptr;        // pinned copy of its abort word, round count and statistics (24 words)
    bool pipe_disabled = false;        // the handle stays on one launch per kernel: three pipeline launches had to be abandoned (see k_pipeline), or the XCD census failed
    int pipe_aborts = 0;               // pipeline launches of this handle abandoned so far (a solve whose launch is abandoned starts over with one launch per kernel)
    static constexpr int PIPE_ABORTS_MAX = 3;
    int last_mode = 0;                 // 0: one launch per kernel and iteration, 1: single-launch pipeline (+ k_solve_wg behind it), 2: k_solve_wg alone
    int32_t* d_counter = nullptr;      // [MAX_GROUPS][MAX_POLL_IT] instances still running after iteration it
    int32_t* h_counter = nullptr;      // pinned, [MAX_GROUPS][2] (double-buffered per poll)
    hipEvent_t ev_poll[4][2] = {};
    // staging buffers of the host entry point
    size_t cap_io = 0;
    double *d_x0 = nullptr, *d_p = nullptr, *d_xout = nullptr, *d_kkt = nullptr, *d_obst = nullptr;
    int32_t *d_status = nullptr, *d_iters = nullptr;
    hipStream_t own_stream = nullptr;
    // sub-batch pipelining: up to MAX_GROUPS tile groups iterate on their own streams so that the latency-bound
    // Riccati sweep of one group overlaps the stage kernels of the others
    static constexpr int MAX_GROUPS = 4;
    hipStream_t sub_stream[MAX_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
    // profiling
    bool profiling = false;
    bool prof_span = false;             // (mpc_set_profiling(h, 2): the iteration loop of the hybrid solve as ONE span -- no marker between its two kernels)
    double prof[6] = {0, 0, 0, 0, 0, 0};
    double res_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // k_solve_wg: ms (profiling only), ran, rounds of the slowest workgroup, workgroups, workgroup-rounds, Riccati sweeps, instance-iterations
    double pipe_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // k_pipeline: ms (profiling only), launches, rounds, Riccati-wait / stage-wait / stage-busy ms summed over workers, items, stage workers + riccati workers / 1000
    int n_cu = 256;
    uint32_t xcd_mask = 0xFFu;          // XCDs seen by k_xcd_census
    std::vector<hipEvent_t> ev_pool;
    uint32_t* d_fail = nullptr;         // [0] instances of the last solve that did not converge (counted by k_egest); [1] sticky abort word of an asynchronous closed loop; [2..5] k_solve_wg: rounds of the slowest workgroup, workgroup-rounds, Riccati sweeps, instance-iterations
    bool async_loop = false;            // solves are being enqueued by the closed-loop driver without host synchronisation
    bool async_ok = false;              // ... and the last one really went out that way
    uint32_t* h_fail = nullptr;         // pinned copy
    int loop_replayed = 0;              // the last closed loop had to be replayed with host synchronisation per step
    int rescued_last = 0;               // instances the last solve handed to the second chance (rescue_dev)
    bool in_rescue = false;             // rescue_dev is solving its levels: they get no second chance of their own
    bool resc_in_kernel = false;        // the last solve ran k_solve_wg with the second chance inside (RESC) over EVERY instance of the batch: rescue_dev has nothing to add
    bool resc_ran = false;              // ... or at least over the instances the pipeline handed over (what stalled inside the pipeline is rescue_dev's)
    bool resc_hint = false;             // the last solve of this handle had stalled instances: the next one runs k_solve_wg<.., RESC> (option rescue_wg = 1, the default)
    bool attr_set_fq = false;
    bool attr_set = false;              // dynamic-LDS limits of the kernels raised on this handle's device
    // run-time switches: read from the environment ONCE, at mpc_create (MPCGPU_<NAME>), changed afterwards only through
    // mpc_set_option -- no getenv on the solve path
    struct Knobs {
        int pipeline = 1, hybrid = 1, hybrid_bx = 0, hybrid_live = -1, pipe_help = -1, pipe_test_abort = 0;
        int rescue = 1, rescue_wg = 1, rescue_alone = 0, loop_async = 1, sync_spin = 1, max_batch = 0, friction_lb = 0, bound_mask = 1, big_wg = 0, groups = 0, poison = 0;
        int timing = 0;                 // profiling aids, a sum of TIMING_* (below): shader-clock stamps printed to stderr; each synchronises
        uint32_t pipe_xcd_mask = 0;
    } knobs;
    // grow-only device scratch of the entry points around the solve (plant step, metrics, FORCES mode): slot -> buffer
    static constexpr int N_SCRATCH = 40;
    void* scratch[N_SCRATCH] = {};
    size_t scratch_cap[N_SCRATCH] = {};
};

// option timing: which kernels stamp their phases with the shader clock (printed to stderr after the solve)
constexpr int TIMING_STAGE = 1,     // k_stage / k_riccati of the third / fourth iteration (forces one launch per kernel)
              TIMING_PIPE = 2,      // k_pipeline: sixth work item / pass of every worker
              TIMING_WG = 4,        // k_solve_wg: third round of every workgroup
              TIMING_START = 8,     // k_start: every workgroup
              TIMING_WG_TRACE = 16; // k_solve_wg: start, end and rounds of every workgroup (100 MHz wall clock)

// error text of the last failed mpc_create on this thread (a handle does not exist yet to carry it)
static thread_local std::string g_create_error;

static int set_knob(mpc_handle::Knobs& k, const char* name, const char* value) {
    const std::string n(name ? name : "");
    const char* v = value ? value : "";
    const long iv = strtol(v, nullptr, 0);
    const int on = (value == nullptr) ? 0 : ((v[0] == '\0') ? 1 : (int)iv);   // "" (variable set, no value) counts as on
    const int on1 = value == nullptr ? 1 : (v[0] != '0');                     // (switches that default to on: unset = on)
    if (n == "pipeline") k.pipeline = on1;
    else if (n == "hybrid") k.hybrid = value == nullptr ? 1 : (int)iv;
    else if (n == "hybrid_bx") k.hybrid_bx = value == nullptr ? 0 : (int)iv;
    else if (n == "hybrid_live") k.hybrid_live = value == nullptr ? -1 : (int)iv;
    else if (n == "pipe_help") k.pipe_help = value == nullptr ? -1 : (int)iv;
    else if (n == "pipe_test_abort") k.pipe_test_abort = on != 0;
    else if (n == "pipe_xcd_mask") k.pipe_xcd_mask = value == nullptr ? 0u : (uint32_t)strtoul(v, nullptr, 0);
    else if (n == "rescue") k.rescue = on1;
    else if (n == "rescue_wg") k.rescue_wg = value == nullptr ? 1 : (int)iv;
    else if (n == "rescue_alone") k.rescue_alone = on != 0;
    else if (n == "loop_async") k.loop_async = on1;
    else if (n == "sync_spin") k.sync_spin = on1;
    else if (n == "max_batch") k.max_batch = value == nullptr ? 0 : (int)iv;
    else if (n == "friction_lb") k.friction_lb = value == nullptr ? 0 : ((std::string(v) == "ipopt") ? 1 : (std::string(v) == "nlp") ? 0 : (int)iv);
    else if (n == "bound_mask") k.bound_mask = value == nullptr ? 1 : (int)iv;
    else if (n == "big_wg") k.big_wg = on != 0;
    else if (n == "groups") k.groups = (int)iv;
    else if (n == "poison") k.poison = on != 0;
    else if (n == "timing") k.timing = value == nullptr ? 0 : (int)iv;
    else return MPC_ERR_INVALID;
    return MPC_OK;
}
static int get_knob(const mpc_handle::Knobs& k, const char* name, long* out) {
    const std::string n(name ? name : "");
    if (n == "pipeline") *out = k.pipeline;
    else if (n == "hybrid") *out = k.hybrid;
    else if (n == "hybrid_bx") *out = k.hybrid_bx;
    else if (n == "hybrid_live") *out = k.hybrid_live;
    else if (n == "pipe_help") *out = k.pipe_help;
    else if (n == "pipe_test_abort") *out = k.pipe_test_abort;
    else if (n == "pipe_xcd_mask") *out = (long)k.pipe_xcd_mask;
    else if (n == "rescue") *out = k.rescue;
    else if (n == "rescue_wg") *out = k.rescue_wg;
    else if (n == "rescue_alone") *out = k.rescue_alone;
    else if (n == "loop_async") *out = k.loop_async;
    else if (n == "sync_spin") *out = k.sync_spin;
    else if (n == "max_batch") *out = k.max_batch;
    else if (n == "friction_lb") *out = k.friction_lb;
    else if (n == "bound_mask") *out = k.bound_mask;
    else if (n == "big_wg") *out = k.big_wg;
    else if (n == "groups") *out = k.groups;
    else if (n == "poison") *out = k.poison;
    else if (n == "timing") *out = k.timing;
    else return MPC_ERR_INVALID;
    return MPC_OK;
}
static void knobs_from_env(mpc_handle::Knobs& k) {
    static const char* names[] = {"pipeline", "hybrid", "hybrid_bx", "hybrid_live", "pipe_help", "pipe_test_abort", "pipe_xcd_mask", "rescue", "rescue_wg", "rescue_alone", "loop_async",
                                  "sync_spin", "max_batch", "friction_lb", "bound_mask", "big_wg", "groups", "poison", "timing"};
    for (const char* n : names) {
        std::string env = "MPCGPU_";
        for (const char* c = n; *c; ++c) env += (char)toupper(*c);
        const char* v = getenv(env.c_str());
        if (v) (void)set_knob(k, n, v);
    }
}
// device scratch buffer `slot` of at least `bytes` (grow-only, owned by the handle)
static void* scratch_get(mpc_handle* h, int slot, size_t bytes) {
    if (bytes == 0) bytes = 8;
    if (h->scratch_cap[slot] < bytes) {
        if (h->scratch[slot]) (void)hipFree(h->scratch[slot]);
        h->scratch[slot] = nullptr; h->scratch_cap[slot] = 0;
        if (hipMalloc(&h->scratch[slot], bytes) != hipSuccess) return nullptr;
        h->scratch_cap[slot] = bytes;
    }
    return h->scratch[slot];
}
// device buffer freed when it goes out of scope (debug / trace buffers of one call)
struct DevTmp {
    void* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

#define HIP_TRY(h, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            char buf_[512];                                                                           \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            (h)->err = buf_;                                                                          \
            return MPC_ERR_HIP;                                                                       \
        }                                                                                             \
    } while (0)

static void free_ws(mpc_handle* h) {
    if (h->d_ws) (void)hipFree(h->d_ws);
    if (h->d_iws) (void)hipFree(h->d_iws);
    h->d_ws = nullptr; h->d_iws = nullptr; h->cap_Bp = 0; h->ws_mailbox = false;
}
static void free_io(mpc_handle* h) {
    void* ptrs[] = {h->d_x0, h->d_p, h->d_xout, h->d_kkt, h->d_obst, h->d_status, h->d_iters};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    h->d_x0 = h->d_p = h->d_xout = h->d_kkt = h->d_obst = nullptr;
    h->d_status = h->d_iters = nullptr;
    h->cap_io = 0;
}

extern "C" {

int mpc_abi_version(void) { return MPCGPU_ABI_VERSION; }

void mpc_default_desc(mpc_problem_desc* desc, int32_t N, int32_t nx) { default_desc(desc, N, nx); }

const char* mpc_last_error(const mpc_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mpc_create(mpc_handle** out, const mpc_problem_desc* desc) {
    if (!out || !desc) { g_create_error = "null argument"; return MPC_ERR_INVALID; }
    *out = nullptr;
    std::string err;
    int rc = validate_desc(*desc, err);
    if (rc) { g_create_error = err; return rc; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e) + " (this library has no CPU path)";
        return MPC_ERR_HIP;
    }
    if (desc->device < 0 || desc->device >= ndev) { g_create_error = "device ordinal out of range"; return MPC_ERR_INVALID; }
    mpc_handle* h = new (std::nothrow) mpc_handle();
    if (!h) { g_create_error = "out of host memory"; return MPC_ERR_INVALID; }
    h->hp.desc = *desc;
    h->device = desc->device;
    knobs_from_env(h->knobs);
    h->hp.fric_literal = h->knobs.friction_lb ? 1 : 0;
    if (hipSetDevice(h->device) != hipSuccess || hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&h->d_counter, sizeof(int32_t) * mpc_handle::MAX_GROUPS * mpc_handle::MAX_POLL_IT) != hipSuccess ||
        hipHostMalloc(&h->h_counter, sizeof(int32_t) * mpc_handle::MAX_GROUPS * 2) != hipSuccess ||
        hipMalloc(&h->d_fail, 8 * sizeof(uint32_t)) != hipSuccess || hipHostMalloc(&h->h_fail, 8 * sizeof(uint32_t)) != hipSuccess) {
        g_create_error = "HIP stream/counter allocation failed";
        delete h;
        return MPC_ERR_HIP;
    }
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && ncu > 0) h->n_cu = ncu;
    }
    {
        // the pipeline routes a tile's work by the XCD a workgroup really runs on: learn the set once (d_counter[0] as scratch)
        // (memset, kernel and read-back all on the handle's own stream: that stream does not synchronise with the null stream, a
        //  hipMemset there could land after the kernel's atomics and leave an empty mask)
        uint32_t m = 0;
        if (hipMemsetAsync(h->d_counter, 0, sizeof(uint32_t), h->own_stream) == hipSuccess) {
            hipLaunchKernelGGL(k_xcd_census, dim3(4 * h->n_cu), dim3(64), 0, h->own_stream, reinterpret_cast<uint32_t*>(h->d_counter));
            if (hipMemcpyAsync(h->h_counter, h->d_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, h->own_stream) == hipSuccess &&
                hipStreamSynchronize(h->own_stream) == hipSuccess && (m = (uint32_t)h->h_counter[0]) != 0u && __builtin_popcount(m) <= 8) h->xcd_mask = m;
            else h->pipe_disabled = true;
        }
    }
    bool ok_streams = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int g = 0; g < mpc_handle::MAX_GROUPS; ++g)
        for (int q = 0; q < 2; ++q) ok_streams = ok_streams && hipEventCreateWithFlags(&h->ev_poll[g][q], hipEventDisableTiming) == hipSuccess;
    for (int g = 0; g < mpc_handle::MAX_GROUPS && ok_streams; ++g)
        ok_streams = hipStreamCreateWithFlags(&h->sub_stream[g], hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&h->ev_join[g], hipEventDisableTiming) == hipSuccess;
    if (!ok_streams) { g_create_error = "HIP stream/event creation failed"; mpc_destroy(h); return MPC_ERR_HIP; }
    rc = mpc_set_bounds(h, nullptr, nullptr, nullptr, nullptr);     // reference defaults until told otherwise
    if (rc) { g_create_error = h->err; mpc_destroy(h); return rc; }
    *out = h;
    return MPC_OK;
}

int mpc_destroy(mpc_handle* h) {
    if (!h) return MPC_OK;
    (void)hipSetDevice(h->device);
    free_ws(h);
    free_io(h);
    if (h->d_LB) (void)hipFree(h->d_LB);
    if (h->d_UB) (void)hipFree(h->d_UB);
    if (h->d_counter) (void)hipFree(h->d_counter);
    if (h->d_pipe) (void)hipFree(h->d_pipe);
    if (h->h_pipe) (void)hipHostFree(h->h_pipe);
    if (h->d_tile_mask) (void)hipFree(h->d_tile_mask);
    if (h->d_state) (void)hipFree(h->d_state);
    for (void* sp : h->scratch) if (sp) (void)hipFree(sp);
    if (h->h_counter) (void)hipHostFree(h->h_counter);
    if (h->d_fail) (void)hipFree(h->d_fail);
    if (h->h_fail) (void)hipHostFree(h->h_fail);
    for (hipEvent_t ev : h->ev_pool) (void)hipEventDestroy(ev);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (int g = 0; g < mpc_handle::MAX_GROUPS; ++g)
        for (int q = 0; q < 2; ++q) if (h->ev_poll[g][q]) (void)hipEventDestroy(h->ev_poll[g][q]);
    for (int g = 0; g < mpc_handle::MAX_GROUPS; ++g) {
        if (h->ev_join[g]) (void)hipEventDestroy(h->ev_join[g]);
        if (h->sub_stream[g]) (void)hipStreamDestroy(h->sub_stream[g]);
    }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return MPC_OK;
}

int mpc_set_bounds(mpc_handle* h, const double* lbx, const double* ubx, const double* lbg, const double* ubg) {
    if (!h) return MPC_ERR_INVALID;
    int rc = set_bounds(h->hp, lbx, ubx, lbg, ubg, h->err);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = h->hp.LB.size() * sizeof(double);
    if (!h->d_LB) { HIP_TRY(h, hipMalloc(&h->d_LB, n)); HIP_TRY(h, hipMalloc(&h->d_UB, n)); }
    HIP_TRY(h, hipMemcpy(h->d_LB, h->hp.LB.data(), n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_UB, h->hp.UB.data(), n, hipMemcpyHostToDevice));
    return MPC_OK;
}

int mpc_set_profiling(mpc_handle* h, int32_t enable) {
    if (!h) return MPC_ERR_INVALID;
    h->profiling = enable != 0;
    h->prof_span = enable == 2;
    return MPC_OK;
}

int mpc_set_option(mpc_handle* h, const char* name, const char* value) {
    if (!h || !name) return MPC_ERR_INVALID;
    const int rc = set_knob(h->knobs, name, value);
    if (rc) h->err = std::string("unknown option (or a value this build does not support): ") + name;
    else apply_fric_literal(h->hp, h->knobs.friction_lb);
    return rc;
}

int mpc_get_option(const mpc_handle* h, const char* name, int64_t* value) {
    if (!h || !name || !value) return MPC_ERR_INVALID;
    long v = 0;
    // (not a switch, a state: 1 after a pipeline launch of this handle had to be abandoned -- it then stays on one launch per kernel)
    if (std::string(name) == "pipe_disabled") { *value = h->pipe_disabled ? 1 : 0; return MPC_OK; }
    // (a count: pipeline launches of this handle that had to be abandoned -- their solves started over with one launch per kernel)
    if (std::string(name) == "pipe_aborts") { *value = h->pipe_aborts; return MPC_OK; }
    const int rc = get_knob(h->knobs, name, &v);
    if (rc == MPC_OK) *value = (int64_t)v;
    return rc;
}

int mpc_last_rescued(const mpc_handle* h) { return h ? h->rescued_last : MPC_ERR_INVALID; }

int mpc_get_pipeline_profile(const mpc_handle* h, double out[8]) {
    if (!h || !out) return MPC_ERR_INVALID;
    for (int i = 0; i < 8; ++i) out[i] = h->pipe_prof[i];
    return MPC_OK;
}

int mpc_get_resident_profile(const mpc_handle* h, double out[8]) {
    if (!h || !out) return MPC_ERR_INVALID;
    for (int i = 0; i < 8; ++i) out[i] = h->res_prof[i];
    return MPC_OK;
}

int mpc_measure_copy_bandwidth(mpc_handle* h, size_t bytes, int32_t reps, double* gbs) {
    if (!h || !gbs || reps < 1 || bytes < (1u << 20)) return MPC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = bytes / 16;
    DevTmp a, b;
    HIP_TRY(h, hipMalloc(&a.p, n * 16));
    HIP_TRY(h, hipMalloc(&b.p, n * 16));
    HIP_TRY(h, hipMemsetAsync(a.p, 1, n * 16, h->own_stream));
    hipEvent_t e0, e1;
    HIP_TRY(h, hipEventCreate(&e0));
    HIP_TRY(h, hipEventCreate(&e1));
    const int blocks = h->n_cu * 16;
    double best = 0.0;
    for (int r = 0; r < reps + 1; ++r) {
        HIP_TRY(h, hipEventRecord(e0, h->own_stream));
        hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, h->own_stream, reinterpret_cast<const uint4*>(a.p), reinterpret_cast<uint4*>(b.p), n);
        HIP_TRY(h, hipEventRecord(e1, h->own_stream));
        HIP_TRY(h, hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms > 0.f) best = std::max(best, 2.0 * (double)n * 16.0 / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *gbs = best;
    return MPC_OK;
}

int mpc_get_profile(const mpc_handle* h, double out[6]) {
    if (!h || !out) return MPC_ERR_INVALID;
    for (int i = 0; i < 6; ++i) out[i] = h->prof[i];
    return MPC_OK;
}

}  // extern "C"

// can this handle, as its options stand, ever launch k_solve_wg (the only user of the instance-major mailbox section of the workspace)?
static bool wants_mailbox(const mpc_handle* h) {
    const mpc_problem_desc& d = h->hp.desc;
    const mpc_handle::Knobs& kn = h->knobs;
    const bool small_wg = 4 * (d.N + 1) <= 256 && !kn.big_wg;
    return small_wg && d.N + 1 <= 64 && kn.hybrid && d.fixed_iters <= 0 && kn.pipeline && !h->pipe_disabled;
}
static int ensure_ws(mpc_handle* h, size_t Bp) {
    const bool mb = wants_mailbox(h);
    if (Bp <= h->cap_Bp && (!mb || h->ws_mailbox)) return MPC_OK;
    const size_t Bp_req = Bp;
    Bp = std::max(Bp, h->cap_Bp);                      // (grow only: also when all that changes is the mailbox section)
    free_ws(h);
    WsLayout w = ws_layout(h->hp.desc.N, h->hp.desc.nx, Bp, mb);
    // (a handle that grew close to the limit WITHOUT the mailbox section and now needs one: what was asked for fits -- max_rows_per_solve
    //  has chunked the batch for the layout with the mailbox -- the old capacity with the section added may not)
    if (w.total * sizeof(double) >= ((size_t)1 << 32) && Bp_req < Bp) { Bp = Bp_req; w = ws_layout(h->hp.desc.N, h->hp.desc.nx, Bp, mb); }
    if (w.total * sizeof(double) >= ((size_t)1 << 32)) {
        h->err = "batch too large: the workspace must stay below 4 GiB (32-bit buffer offsets); split the batch";
        return MPC_ERR_INVALID;
    }
    HIP_TRY(h, hipMalloc(&h->d_ws, w.total * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->d_iws, w.itotal * sizeof(int32_t)));
    h->cap_Bp = Bp;
    h->ws_mailbox = mb;
    return MPC_OK;
}

namespace {
struct Prof {
    mpc_handle* h;
    hipStream_t s;
    size_t used = 0;
    std::vector<int> kinds;     // kind per (start, stop) pair
    std::vector<std::pair<size_t, size_t>> spans;      // the pair's events (indices into the pool)
    bool open = false;
    hipEvent_t get() {
        if (used == h->ev_pool.size()) {
            hipEvent_t ev;
            (void)hipEventCreate(&ev);
            h->ev_pool.push_back(ev);
        }
        return h->ev_pool[used++];
    }
    void begin(int kind, hipStream_t st = nullptr) {
        if (!h->profiling) return;
        if (open) { next(kind, st); return; }          // (a span left open on purpose: its end is this one's start)
        kinds.push_back(kind);
        (void)hipEventRecord(get(), st ? st : s);
        spans.push_back({used - 1, 0});
        open = true;
    }
    void end(hipStream_t st = nullptr) {
        if (!h->profiling || !open) return;
        (void)hipEventRecord(get(), st ? st : s);
        spans.back().second = used - 1;
        open = false;
    }
    // end of one span = start of the next, ONE event (a marker between two kernels costs the stream a microsecond or two: the kernels of the
    // iteration loop are measured back to back with a single marker between them)
    void next(int kind, hipStream_t st = nullptr) {
        if (!h->profiling) return;
        if (!open) {
            kinds.push_back(kind);
            (void)hipEventRecord(get(), st ? st : s);
            spans.push_back({used - 1, 0});
            open = true;
            return;
        }
        (void)hipEventRecord(get(), st ? st : s);
        spans.back().second = used - 1;
        kinds.push_back(kind);
        spans.push_back({used - 1, 0});
    }
    void collect() {
        for (int i = 0; i < 6; ++i) h->prof[i] = 0;
        if (!h->profiling) return;
        (void)hipDeviceSynchronize();
        for (size_t i = 0; i < kinds.size(); ++i) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, h->ev_pool[spans[i].first], h->ev_pool[spans[i].second]);
            if (kinds[i] == 0) { h->prof[0] += ms; h->prof[1] += 1; }
            else if (kinds[i] == 1) { h->prof[2] += ms; h->prof[3] += 1; }
            else if (kinds[i] == 3) h->pipe_prof[0] += ms;
            else if (kinds[i] == 5) h->res_prof[0] += ms;
            else h->prof[4] += ms;
        }
    }
};
}  // namespace

// the one synchronisation of a pipelined solve: polling the stream keeps the host thread on the core for the ~1 ms the launch lasts
// and saves the wake-up latency of a blocking wait (tens of microseconds per solve); option "sync_spin" = 0 blocks instead
static hipError_t wait_stream(mpc_handle* h, hipStream_t stream) {
    if (h->knobs.sync_spin) {
        for (int spins = 0; spins < 2000000; ++spins) {
            const hipError_t e = hipStreamQuery(stream);
            if (e != hipErrorNotReady) return e;
        }
    }
    return hipStreamSynchronize(stream);
}

template <int NX>
static int solve_dev_impl(mpc_handle* h, int32_t B, const double* d_x0, const double* d_p, const double* d_obst,
                          double* d_x_out, int32_t* d_status, int32_t* d_iters, double* d_kkt, hipStream_t stream,
                          double* trace, int32_t trace_rows, int32_t* n_it_out) {
    const mpc_problem_desc& d = h->hp.desc;
    const size_t Bp = ((size_t)B + 63) / 64 * 64;
    // tile-major layout: the rows of a tile do not depend on the batch size, so a smaller batch lives in the first tiles of
    // a larger allocation (grow-only; the rescue path alternates between the full batch and a failed subset)
    int rc = ensure_ws(h, Bp);
    if (rc) return rc;
    // 256-thread stage workgroups (one wave per SIMD, the whole register file, no scratch) hold 8 instances up to N = 31 and
    // 4 up to N = 63 (N = 50, B = 4096: 2.49 ms against 3.25 ms with 512-thread workgroups, 1.99 ms in the pipeline)
    const mpc_handle::Knobs& kn = h->knobs;
    const bool small_wg = 4 * (d.N + 1) <= 256 && !kn.big_wg;
    int bx = pick_bx(d.N, small_wg ? 256 : STAGE_MAX_THREADS);
    Params P;
    P.mbw_live = 0;
    fill_params(P, h->hp, B, Bp, bx, h->d_ws, h->d_iws, h->d_LB, h->d_UB, h->ws_mailbox);
    P.x0 = d_x0; P.p = d_p; P.x_out = d_x_out; P.status_out = d_status; P.iters_out = d_iters; P.kkt_out = d_kkt;
    const WsLayout w = ws_layout(d.N, d.nx, Bp, h->ws_mailbox);
    Prof prof{h, stream};
    h->async_ok = false;
    if (d_obst) {
        P.per_inst_obst = 1;
        hipLaunchKernelGGL(k_transpose_obst, dim3((B + 255) / 256), dim3(256), 0, stream, d_obst, h->d_ws + w.OBST * 64, B, (uint32_t)w.tile_elems);
    }
    const int S = d.N + 1;
    const int threads = ((S * bx + 63) / 64) * 64;
    const bool stage_timing = (kn.timing & TIMING_STAGE) != 0;
    const int nblk = (B + bx - 1) / bx;
    const int nw = threads / 64;
    // LDS: reductions | the larger of (stage exchange, multiplier stash of the 256-thread variant) | prefetch images
    const bool has_ou = h->hp.has_ou != 0;
    const size_t lds_max = 160 * 1024 - 1024;          // the kernels also hold a few hundred bytes of static LDS
    const int stash_rows = small_wg && MPC_STAGE_STASH ? std::max(Stash<NX>::rows(has_ou), 2 * NX) : 2 * NX;
    const size_t lds_bytes = ((size_t)nw * 10 * bx + (size_t)2 * S * (NX + 2) + (size_t)stash_rows * threads) * sizeof(double);
    // the start-iterate kernel parks nothing in LDS (no line search): with the small footprint several of its workgroups share a CU
    // and the 512 of a 4096-instance batch run at once instead of in two rounds
    const size_t lds_init = ((size_t)nw * 10 * bx + (size_t)2 * S * (NX + 2) + (size_t)2 * NX * threads) * sizeof(double);
    const int rblk = (int)(Bp / 64);
    const size_t ric_lds = std::max(RIC_DEPTH * (size_t)((MPC_EV(Dim<NX>::NBLK) * 512 + 1023) / 1024) * 1024,
                                    RIC_DEPTH_F * (size_t)((Dim<NX>::NKK * 512 + 1023) / 1024 + 6) * 1024) + 64;   // ring + flag
    {
        if (!h->attr_set) {        // per handle: the attribute belongs to the function object of the handle's device
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_riccati<NX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ric_lds));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stage<NX, false, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_start<NX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prestart_par<NX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stage<NX, true, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stage<NX, false, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stage<NX, true, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pipeline<NX, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_wg<NX, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stage<NX, false, 256, REF_VM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stage<NX, false, 512, REF_VM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pipeline<NX, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pipeline<NX, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_wg<NX, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_wg<NX, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_wg<NX, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            h->attr_set = true;
        }
    }

    // ---- tile groups (sub-batches).  Every group runs ingest -> init -> iterations -> egest on its own stream.
    const int ntiles = (int)(Bp / 64);
    int G = 1;
    // (measured on MI355X at B = 4096: 4 groups gain 4 % in fixed-iteration mode and lose in converged mode, where
    //  every group polls on its own; the workgroups of the two kernels cannot share a CU, so the overlap is small.
    //  Kept as an opt-in: MPCGPU_GROUPS=2..4.)
    if (!trace && !stage_timing && kn.groups > 0) {
        G = kn.groups;
        if (G > ntiles / 4) G = ntiles / 4;
        if (G < 1) G = 1;
        if (G > mpc_handle::MAX_GROUPS) G = mpc_handle::MAX_GROUPS;
    }
    const bool ref_vm_ok = ((P.lo_mask | P.hi_mask) & ~REF_BOUND_VM) == 0u;    // bounds on the reference's four variables only
    // one thread per stage with the reference's bound structure compiled in (option bound_mask, default on): same arithmetic, same bits
    const bool masked = kn.bound_mask != 0 && ref_vm_ok && (P.dense_mask & REF_DENSE_LO) == REF_DENSE_LO && ((P.dense_mask >> 8) & REF_DENSE_HI) == REF_DENSE_HI && P.has_ol && !P.has_ou && P.obst_mult == 3 && !P.per_inst_obst;
    struct Group { int tile0, ntl, blk0, nblk, b0, b1; hipStream_t st; bool running; };
    Group grp[mpc_handle::MAX_GROUPS];
    for (int g = 0; g < G; ++g) {
        Group& q = grp[g];
        q.tile0 = (int)((long long)ntiles * g / G);
        q.ntl = (int)((long long)ntiles * (g + 1) / G) - q.tile0;
        q.b0 = q.tile0 * 64;
        q.b1 = std::min(B, (q.tile0 + q.ntl) * 64);
        q.nblk = (q.b1 - q.b0 + bx - 1) / bx;
        q.st = (G == 1) ? stream : h->sub_stream[g];
        q.running = q.b1 > q.b0;
    }
    if (G > 1) {
        HIP_TRY(h, hipEventRecord(h->ev_fork, stream));
        for (int g = 0; g < G; ++g) HIP_TRY(h, hipStreamWaitEvent(h->sub_stream[g], h->ev_fork, 0));
    }
    auto launch_stage = [&](const Group& q, bool init) {
        Params Pg = P;
        Pg.tile0 = q.tile0;
        if (!init && stage_timing && P.DBG) Pg.DBG = P.DBG;
        if (small_wg) {
            if (init) hipLaunchKernelGGL((k_stage<NX, true, 256>), dim3(q.nblk), dim3(threads), lds_init, q.st, Pg, h->hp.n_mult, h->hp.n_z, stash_rows);
            else if (masked) hipLaunchKernelGGL((k_stage<NX, false, 256, REF_VM>), dim3(q.nblk), dim3(threads), lds_bytes, q.st, Pg, h->hp.n_mult, h->hp.n_z, stash_rows);
            else hipLaunchKernelGGL((k_stage<NX, false, 256>), dim3(q.nblk), dim3(threads), lds_bytes, q.st, Pg, h->hp.n_mult, h->hp.n_z, stash_rows);
        } else {
            if (init) hipLaunchKernelGGL((k_stage<NX, true, 512>), dim3(q.nblk), dim3(threads), lds_init, q.st, Pg, h->hp.n_mult, h->hp.n_z, stash_rows);
            else if (masked) hipLaunchKernelGGL((k_stage<NX, false, 512, REF_VM>), dim3(q.nblk), dim3(threads), lds_bytes, q.st, Pg, h->hp.n_mult, h->hp.n_z, stash_rows);
            else hipLaunchKernelGGL((k_stage<NX, false, 512>), dim3(q.nblk), dim3(threads), lds_bytes, q.st, Pg, h->hp.n_mult, h->hp.n_z, stash_rows);
        }
    };
    if (kn.poison && !h->in_rescue) {           // (debugging aid: NaN into every row of every tile of the workspace before the solve)
        const WsLayout wl = ws_layout(d.N, d.nx, Bp, h->ws_mailbox);
        hipLaunchKernelGGL(k_poison, dim3(1024), dim3(256), 0, stream, h->d_ws, (uint32_t)wl.tile_elems, (uint32_t)wl.ntiles, 0u, (uint32_t)wl.rows * 64u);
    }
    const int cap = d.fixed_iters > 0 ? d.fixed_iters : d.max_iter;
    const int chunk = d.fixed_iters > 0 ? cap : 4;
    DevTmp t_trace, t_dbg, t_pdbg;
    double* d_trace = nullptr;
    if (trace) { HIP_TRY(h, hipMalloc(&t_trace.p, sizeof(double) * 8 * (size_t)B)); d_trace = t_trace.as<double>(); }
    auto record_trace = [&](int it) -> int {
        if (!trace || it >= trace_rows) return MPC_OK;
        hipLaunchKernelGGL(k_gather_trace, dim3((B + 255) / 256), dim3(256), 0, stream, h->d_ws + w.SC * 64, (uint32_t)w.tile_elems, B, d_trace);
        HIP_TRY(h, hipStreamSynchronize(stream));
        HIP_TRY(h, hipMemcpy(trace + (size_t)it * 8 * B, d_trace, sizeof(double) * 8 * (size_t)B, hipMemcpyDeviceToHost));
        return MPC_OK;
    };
    unsigned long long* d_dbg = nullptr;
    if (stage_timing) {
        HIP_TRY(h, hipMalloc(&t_dbg.p, sizeof(unsigned long long) * 16 * (size_t)nblk));
        d_dbg = t_dbg.as<unsigned long long>();
        HIP_TRY(h, hipMemsetAsync(d_dbg, 0, sizeof(unsigned long long) * 16 * (size_t)nblk, stream));
    }
    // Convergence polling: every stage launch adds the number of instances it leaves running to its own device counter;
    // after each chunk the last counter is copied to pinned memory.  The host looks at the poll of chunk c-1 only after
    // chunk c is enqueued, so the GPU never idles on a host round trip; the price is one chunk of early-exit launches
    // (a few microseconds each) at the very end.
    if (h->tile_mask_cap < (size_t)ntiles) {
        if (h->d_tile_mask) (void)hipFree(h->d_tile_mask);
        h->d_tile_mask = nullptr;
        HIP_TRY(h, hipMalloc(&h->d_tile_mask, sizeof(unsigned long long) * (size_t)ntiles));
        h->tile_mask_cap = (size_t)ntiles;
    }
    P.tile_mask = h->d_tile_mask;
    const bool polled = d.fixed_iters <= 0 && !trace;
    if (polled && cap > mpc_handle::MAX_POLL_IT) { h->err = "max_iter exceeds the poll table (1024)"; return MPC_ERR_INVALID; }
    int it = 0, chunk_id = 0;
    // ---- single-launch pipeline (k_pipeline): all iterations in one persistent grid, tiles cycling independently.
    // Pays while the Riccati chain is latency bound (few tiles per CU); larger batches keep one launch per kernel.
    h->last_mode = 0;
    for (int i = 0; i < 8; ++i) { h->pipe_prof[i] = 0; h->res_prof[i] = 0; }
    bool piped = false;
    // k_solve_wg with `bxw` instances per workgroup (1 or 2: one wavefront per workgroup, four workgroups per CU; bx: a whole CU)
    // LDS of a k_solve_wg workgroup (ONE wavefront, bxw = 1 or 2 instances): records + bounds table; a restart of the second chance runs the
    // start-point safeguard and the start iterate in the same memory
    auto wg_lds = [&](int bxw) {
        const size_t pre = prestart_doubles(NX, S, bxw) * sizeof(double);
        const size_t init = ((size_t)10 * bxw + (size_t)2 * S * (NX + 2) + (size_t)2 * NX * 64) * sizeof(double);
        return std::max(WgLds<NX>::doubles(S, bxw) * sizeof(double), bxw == 1 ? std::max(pre, init) : (size_t)0);
    };
    // the second chance inside the launch (k_solve_wg<.., RESC>): one instance per workgroup, the conditions of rescue_dev
    // (option rescue_wg: 0 never; 2 always; 1, the default: when the handle's LAST solve had stalled instances -- the kernel with the second chance
    //  inside carries its restart code at 512 registers and ~1 KB of scratch per thread, which costs a batch that never stalls 6 - 10 %: B = 256 lane
    //  following 0.405 -> 0.366 ms, N = 50 1.24 -> 1.17 ms; the two give the same bits, so a handle may change between them from solve to solve)
    const bool resc_cond = kn.rescue && (kn.rescue_wg >= 2 || kn.rescue_alone || (kn.rescue_wg == 1 && h->resc_hint)) && d.fixed_iters <= 0 && !trace && h->hp.has_ol && h->hp.ol_raw > 0.0 && !h->in_rescue;
    auto wg_resc = [&](int bxw) { return resc_cond && bxw == 1; };
    DevTmp t_wtrace;
    unsigned long long* d_wtrace = nullptr;
    int n_wtrace = 0;
    const int32_t* wg_list = nullptr;                   // (behind the pipeline: the instances its retiring tiles left, see k_solve_wg)
    const uint32_t* wg_list_n = nullptr;
    int wg_grid = 0;
    auto launch_wg = [&](int bxw, const uint32_t* skip_if, uint32_t* stats, uint32_t* fin_ctl) {
        if ((kn.timing & TIMING_WG_TRACE) && !h->async_loop && !h->in_rescue) {
            n_wtrace = (B + bxw - 1) / bxw;
            if (hipMalloc(&t_wtrace.p, sizeof(unsigned long long) * 4 * (size_t)n_wtrace) == hipSuccess) {
                d_wtrace = t_wtrace.as<unsigned long long>();
                (void)hipMemsetAsync(d_wtrace, 0, sizeof(unsigned long long) * 4 * (size_t)n_wtrace, stream);
            }
        }
        Params Pw = P;
        Pw.bx = bxw;
        Pw.fin_ctl = fin_ctl; Pw.fin_host = h->h_pipe;
        if (skip_if != nullptr) Pw.DBG = nullptr;         // (behind the pipeline: a stamp buffer of option pipe_timing is sized for the PIPELINE's workgroups)
        const int thr = 64;                        // (S * bxw <= 64: checked where the path is chosen)
        WgRescue rs{h->hp.ol_raw, BOUND_RELAX, 0};
        const dim3 grid(wg_grid > 0 ? wg_grid : (B + bxw - 1) / bxw);
        if (wg_resc(bxw)) {
            rs.on = 1;
            h->resc_ran = true;
            // (behind the pipeline the launch sees the instances on the hand-over lists only: one that stalled INSIDE the pipeline is on none of
            //  them and keeps its status for rescue_dev)
            h->resc_in_kernel = wg_list == nullptr;
            if (masked) hipLaunchKernelGGL((k_solve_wg<NX, 2, true>), grid, dim3(thr), wg_lds(bxw), stream, Pw, h->hp.n_mult, h->hp.n_z, stash_rows, stats, skip_if, rs, d_wtrace, wg_list, wg_list_n);
            else hipLaunchKernelGGL((k_solve_wg<NX, 0, true>), grid, dim3(thr), wg_lds(bxw), stream, Pw, h->hp.n_mult, h->hp.n_z, stash_rows, stats, skip_if, rs, d_wtrace, wg_list, wg_list_n);
        }
        else if (masked) hipLaunchKernelGGL((k_solve_wg<NX, 2>), grid, dim3(thr), wg_lds(bxw), stream, Pw, h->hp.n_mult, h->hp.n_z, stash_rows, stats, skip_if, rs, d_wtrace, wg_list, wg_list_n);
        else hipLaunchKernelGGL((k_solve_wg<NX, false>), grid, dim3(thr), wg_lds(bxw), stream, Pw, h->hp.n_mult, h->hp.n_z, stash_rows, stats, skip_if, rs, d_wtrace, wg_list, wg_list_n);
    };
    // option wg_trace: every workgroup of k_solve_wg leaves its start, its end (100 MHz wall clock) and its rounds: when did the long ones start?
    auto report_wtrace = [&]() {
        if (!d_wtrace) return;
        std::vector<unsigned long long> hw((size_t)4 * n_wtrace);
        if (hipMemcpy(hw.data(), d_wtrace, hw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
        unsigned long long t0 = ~0ull, t1 = 0ull;
        std::vector<int> live;
        for (int w = 0; w < n_wtrace; ++w) if (hw[4 * w + 2]) { t0 = std::min(t0, hw[4 * w]); t1 = std::max(t1, hw[4 * w + 1]); live.push_back(w); }
        if (live.empty()) return;
        std::sort(live.begin(), live.end(), [&](int a, int b) { return hw[4 * a + 1] > hw[4 * b + 1]; });
        int late = 0;
        double start_max = 0;
        for (int w : live) { const double st = (double)(hw[4 * w] - t0) * 1e-2; if (st > 5.0) ++late; start_max = std::max(start_max, st); }
        fprintf(stderr, "[mpcgpu wg_trace] %d workgroups with work of %d; span %.1f us; %d of them start more than 5 us after the first (latest start %.1f us); the last to finish:\n",
                (int)live.size(), n_wtrace, (double)(t1 - t0) * 1e-2, late, start_max);
        for (size_t i = 0; i < live.size() && i < 12; ++i) {
            const int w = live[i];
            const double st = (double)(hw[4 * w] - t0) * 1e-2, en = (double)(hw[4 * w + 1] - t0) * 1e-2;
            const double tf = (double)((hw[4 * w + 3] >> 16) & 0xFFFFFFull) * 1e-2, tr1 = (double)((hw[4 * w + 3] >> 40) & 0xFFFFFFull) * 1e-2;
            fprintf(stderr, "    workgroup %5d: start %6.1f us  end %6.1f us  rounds %2d  instance-rounds %2d  -> %.1f us per round; instances taken over after %.1f us, first round done after %.1f us, later rounds %.1f us each\n",
                    w, st, en, (int)hw[4 * w + 2], (int)(hw[4 * w + 3] & 0xFFFFu), (en - st) / (double)hw[4 * w + 2], tf, tr1, hw[4 * w + 2] > 1 ? (en - st - tr1) / (double)(hw[4 * w + 2] - 1) : 0.0);
        }
    };
    // hybrid solve (option hybrid): the pipeline runs a tile while it has many instances iterating, then k_solve_wg finishes the
    // stragglers one wavefront per (hybrid_bx) instance -- `hand` = live instances per tile at which a tile changes over
    // instances per wavefront of k_solve_wg: two fill the lanes (62 of 64 at N = 30) and halve the wavefronts a full machine needs; a batch
    // that fits the machine one instance per wavefront (4 wavefront slots per CU) is faster that way -- a round of a wavefront with two live
    // instances costs 47 us against 33 us (B = 256: 0.58 -> 0.54 ms).  hybrid_bx = 1 / 2 pins it, 0 chooses.
    int hyb_bx = ((kn.hybrid_bx == 2 || (kn.hybrid_bx == 0 && B > 4 * h->n_cu)) && S * 2 <= 64) ? 2 : 1;
    // (the levels of the second chance, rescue_dev: always the stragglers' kernel alone with one instance per wavefront -- the configuration the second
    //  chance INSIDE a launch runs in, whatever the number of stalled instances: the pipeline's sweeps round differently from the matrix-pipe sweeps of
    //  k_solve_wg, and which of the two paths gave an instance its second chance must not show in its bits)
    // A handle whose last solve needed second chances: one instance per wavefront also somewhat beyond the machine's slots -- as far as the stragglers'
    // kernel serves such a batch alone anyway, below -- so that the kernel with the second chance inside takes it (collision avoidance B = 1025
    // 3.49 -> 2.83 ms, B = 2304 6.62 -> 4.81 ms; the same bits: an instance's arithmetic in k_solve_wg does not depend on its wavefront's company).
    if (kn.hybrid_bx == 0 && resc_cond && hyb_bx == 2 && (size_t)ntiles * 64 * 8 <= (size_t)4 * h->n_cu * 18) hyb_bx = 1;
    // Option rescue_alone (the caller knows the family stalls -- collision avoidance: a few instances per thousand, whose chains of 50 + 30 ... 60
    // iterations are what the batch waits for): the stragglers' kernel ALONE, one instance per wavefront, with the second chance inside, up to 8192
    // instances -- the bulk of the batch fits beside those chains, while the levels behind a launch cost a synchronised solve each: B = 3072 / 4096
    // 8.3 -> 4.9 ms, 6144 9.2 -> 7.3, 8192 13.7 -> 12.4 ms.  An option and not the handle's history: it changes which sweeps serve an instance (last
    // bits), and two consecutive calls of a handle give the same bits.
    const bool resc_alone = kn.rescue_alone && kn.hybrid_bx == 0 && kn.hybrid_live < 0 && resc_cond && ntiles <= 128;
    if (resc_alone) hyb_bx = 1;
    if (h->in_rescue) hyb_bx = 1;
    // (not with a fixed iteration count: no instance ever stops iterating, so no tile would ever change over)
    const bool hyb_ok = kn.hybrid && h->ws_mailbox && d.fixed_iters <= 0 && kn.pipeline && !h->pipe_disabled && G == 1 && small_wg && S <= 64 && wg_lds(hyb_bx) <= lds_max / 4 && !trace && !stage_timing;
    int hand = 0;
    if (hyb_ok) {
        // what the straggler launch holds at once, per tile (4 wavefront slots per CU x hyb_bx instances) ...
        const int base = std::min(64, 4 * h->n_cu * hyb_bx / std::max(1, (int)(Bp / 64)));
        // ... and half as much again when k_solve_wg deals its workgroups from the hand-over lists: the instances beyond the machine's slots are
        // the ones closest to convergence and start when the first wavefronts free up, while the tiles leave the pipeline a round earlier --
        // its rounds cost an instance 67 us, a straggler round 25 us (tools/hand_sweep.py: 3 - 5 % per batch on four instance sets at B = 4096,
        // B = 3000 / 8192 and N = 50 likewise; above ~50 of 64 the pipeline no longer carries the bulk)
        hand = base >= 64 ? base : std::max(base, std::min(50, base < 32 ? 7 * base / 4 : 3 * base / 2));
        // ... and with ONE instance per wavefront there (horizons beyond 31 stages) at least 40: a round of the pipeline costs such a tile ~100 us, a
        // straggler round 30, and the wavefronts beyond the machine's slots start as the first ones retire (tools/hand_sweep.py n50, four instance
        // sets at B = 4096: 1.17 - 1.21 -> 1.14 - 1.17 ms; B = 5000 / 6000 / 8192: -6 / -10 / -10 %; B = 3000 unchanged)
        if (hyb_bx == 1 && base < 64 && hand < 40) hand = 40;
        // A batch somewhat larger than the stragglers' launch holds at once is still faster there alone than through the pipeline -- the wavefronts
        // beyond the machine's slots start as the first ones retire --: up to 9/8 of the slots with two instances per wavefront (N = 30: B = 2049
        // 0.75 -> 0.57 ms, 2304 0.74 -> 0.72, 2432 already 0.72 against 0.75), up to 9/4 with one, where a round of the pipeline costs three
        // straggler rounds (N = 50: B = 1025 0.83 -> 0.54 ms, 1536 0.92 -> 0.77, 2304 0.98 -> 0.92, 2560 equal)
        if ((size_t)ntiles * 64 * 8 <= (size_t)4 * h->n_cu * hyb_bx * (hyb_bx == 2 ? 9 : 18)) hand = 64;
        if (resc_alone) hand = 64;
        if (kn.hybrid_live >= 0) hand = std::min(64, kn.hybrid_live);
    }
    const bool wg_only = hyb_ok && (hand >= 64 || h->in_rescue);             // every tile would change over at once: no pipeline launch at all
    // ---- which path serves the iteration loop (decided before anything is launched: the kernels of the loop write the caller's rows themselves)
    uint32_t xcd_mask = h->xcd_mask;
    if (kn.pipe_xcd_mask) {            // tests: pretend some XCDs away (a partitioned device); workgroups that land there leave
        const uint32_t m = kn.pipe_xcd_mask & h->xcd_mask;
        if (m) xcd_mask = m;
    }
    const int n_xcd = __builtin_popcount(xcd_mask);
    const int tiles_x = (ntiles + n_xcd - 1) / n_xcd;
    const int cu_x = std::max(2, h->n_cu / n_xcd);                     // a quarter of an XCD's CUs run Riccati sweeps (8 of 32)
    // (nine to twelve tiles per XCD -- batches just above 4096 instances --: every tile its own Riccati worker still; with eight, one worker would own
    //  two tiles and every round of the XCD would wait for its two passes -- as long as the stage workers that remain are not the scarcer kind:
    //  at most five stage items per round for each of them.  tools/size_sweep.py, N = 30: B = 4160 ... 6144 -10 ... -4 %; N = 50, sixteen items per
    //  tile, loses 2 - 11 % with it and keeps eight.)
    const int n_ric_base = std::max(1, cu_x / 4);
    // (... and the sweep of a tile is long against its stage items: at N = 10 a worker's second pass hides behind the first tile's stage work,
    //  and B = 6144 loses 9 % to the four stage workers it gives up)
    const bool ric_per_tile = tiles_x > n_ric_base && tiles_x <= 3 * cu_x / 8 && (64 / bx) * tiles_x <= 5 * (cu_x - tiles_x) && S >= 24;
    const int n_ric = std::min(ric_per_tile ? tiles_x : n_ric_base, tiles_x);
    // (threads >= 192: a stage item then covers at least six stages per wavefront and instance column -- the shapes the hand-off timing was measured on)
    const bool eligible = G == 1 && !trace && !stage_timing && small_wg && threads >= 192 && threads <= 256 &&
                          ntiles <= (d.fixed_iters > 0 ? 64 : PIPE_MAX_TILES) && (tiles_x + n_ric - 1) / n_ric <= 32 &&
                          std::max(lds_bytes, ric_lds) <= lds_max;
    const bool res_path = wg_only;         // k_solve_wg alone
    const bool pipe_path = !res_path && eligible && !h->pipe_disabled && kn.pipeline;
    // Both get a control block (abort word, round count, statistics of both loop kernels, count of instances that did not converge, hand-over
    // lists' counters; the pipeline's queues behind them) -- one of two that alternate: the start kernel of a solve zeroes the other one for the
    // next solve, the last workgroup of the loop copies the head of this one into the handle's pinned block.
    uint32_t *ctl = nullptr, *zero_next = nullptr;
    size_t ctl_words = 0;
    bool next_zeroed = false;
    uint32_t pipe_cap = 1;
    if (res_path || pipe_path) {
        while (pipe_path && pipe_cap < 2u * (64u / (uint32_t)bx) * (uint32_t)tiles_x) pipe_cap <<= 1;
        ctl_words = pipe_ctl_words((uint32_t)ntiles, pipe_cap);
        if (h->pipe_words < ctl_words) {
            if (h->d_pipe) (void)hipFree(h->d_pipe);
            h->d_pipe = nullptr; h->pipe_words = 0;
            HIP_TRY(h, hipMalloc(&h->d_pipe, 2 * ctl_words * sizeof(uint32_t)));
            h->pipe_words = ctl_words;
            h->pipe_clean[0] = h->pipe_clean[1] = 0;
        }
        if (!h->h_pipe) HIP_TRY(h, hipHostMalloc(&h->h_pipe, PIPE_FIN_WORDS * sizeof(uint32_t)));
        ctl = h->d_pipe + (size_t)h->pipe_flip * h->pipe_words;
        zero_next = h->d_pipe + (size_t)(h->pipe_flip ^ 1) * h->pipe_words;
        if (h->pipe_clean[h->pipe_flip] < ctl_words) HIP_TRY(h, hipMemsetAsync(ctl, 0, ctl_words * sizeof(uint32_t), stream));      // (first use, or a larger batch than the last)
        h->pipe_clean[h->pipe_flip] = 0;
        P.emit = 1;
        // (instances that did not converge are counted into word 14 of the block; an asynchronous closed loop accumulates them over its steps in d_fail instead)
        P.fail_count = h->async_loop ? h->d_fail : ctl + PIPE_ABORT + 14;
    }
    // ---- the front of the solve: caller's rows -> workspace, start-point safeguard, start iterate
    for (int g = 0; g < G; ++g) {
        const Group& q = grp[g];
        if (!q.running) continue;
        Params Pg = P;
        Pg.tile0 = q.tile0;
        prof.begin(2, q.st);
        const int n_w = 2 * d.N + NX * (d.N + 1);
        // start-point safeguard: stage-parallel form when its LDS footprint fits the default limit and the horizon has the two
        // stage-threads the scans need (otherwise the two-chain kernel)
        const size_t lds_pre = prestart_doubles(NX, S, bx) * sizeof(double);
        const size_t lds_in = (size_t)bx * (2 * n_w - 2 * d.N) * sizeof(double);             // the block's rows of x0 and of the X_ref part of p
        // (the fused kernel keeps the safeguard's LDS and the block's caller rows side by side; two of its workgroups share a CU)
        const size_t lds_red = (size_t)(threads / 64) * 10 * bx * sizeof(double);            // (stage_block's reduction scratch, in front of the safeguard's region)
        const bool fused = lds_red + lds_pre + 16 + lds_in <= 78 * 1024 && d.N >= 1 && small_wg;
        if (!fused) hipLaunchKernelGGL((k_ingest<NX>), dim3(q.ntl, (n_w + 63) / 64 + (n_w - 2 * d.N + 63) / 64), dim3(256), 0, q.st, Pg);
        if (fused) {
            // (one launch for ingest, safeguard and start iterate: same blocks, same threads)
            DevTmp t_sdbg;
            if ((kn.timing & TIMING_START) && !h->async_loop && G == 1 && hipMalloc(&t_sdbg.p, sizeof(unsigned long long) * 16 * (size_t)q.nblk) == hipSuccess) {
                (void)hipMemsetAsync(t_sdbg.p, 0, sizeof(unsigned long long) * 16 * (size_t)q.nblk, q.st);
                Pg.DBG = t_sdbg.as<unsigned long long>();
            }
            hipLaunchKernelGGL((k_start<NX>), dim3(q.nblk), dim3(threads), std::max(lds_red + lds_pre + 16 + lds_in, lds_init), q.st, Pg, h->hp.n_mult, h->hp.n_z, stash_rows,
                               g == 0 ? zero_next : (uint32_t*)nullptr, (uint32_t)ctl_words);
            if (g == 0 && zero_next != nullptr) next_zeroed = true;
            if (Pg.DBG) {          // (option start_timing: shader-clock stamps of every workgroup of k_start; synchronises)
                std::vector<unsigned long long> hd((size_t)16 * q.nblk);
                if (hipStreamSynchronize(q.st) == hipSuccess && hipMemcpy(hd.data(), Pg.DBG, hd.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                    const int order[16] = {11, 12, 13, 1, 2, 3, 4, 5, 6, 14, 15, 0, 7, 8, 9, 10};
                    static const char* names[15] = {"rows->LDS", "Z/REF stores", "bounds+a0", "defects", "scan1", "tan+scan2", "sincos+scan3", "ROLL+sums", "decide", "fence", "enter", "init point+exchange", "eval+assemble", "reduce", "finish"};
                    double acc[15] = {0};
                    unsigned long long t0 = ~0ull, t1 = 0ull;
                    for (int bq = 0; bq < q.nblk; ++bq) {
                        const unsigned long long* r = hd.data() + (size_t)bq * 16;
                        for (int j = 0; j < 15; ++j) acc[j] += (double)(long long)(r[order[j + 1]] - r[order[j]]);
                        t0 = std::min(t0, r[11]); t1 = std::max(t1, r[10]);
                    }
                    fprintf(stderr, "[mpcgpu k_start timing, shader-clock ticks, mean over %d workgroups]", q.nblk);
                    for (int j = 0; j < 15; ++j) fprintf(stderr, " %s=%.0f", names[j], acc[j] / q.nblk);
                    fprintf(stderr, "; first start to last end %.0f\n", (double)(t1 - t0));
                }
                Pg.DBG = nullptr;
            }
        } else {
            hipLaunchKernelGGL((k_prestart_par<NX>), dim3(q.nblk), dim3(threads), lds_pre, q.st, Pg);       // (N = 127, bx = 4: 112 KB of LDS)
            launch_stage(q, true);
        }
        if (G > 1) prof.end(q.st);          // (one stream: the span stays open, the first kernel of the loop starts where it ends)
    }

    if (ctl != nullptr) {
        h->pipe_clean[h->pipe_flip ^ 1] = next_zeroed ? ctl_words : 0;
        h->pipe_flip ^= 1;
    }
    if (res_path) {
        // ---- workgroup-resident solve alone: ALL iterations of every instance in one launch of k_solve_wg
        const int nblk_dbg = (B + hyb_bx - 1) / hyb_bx;
        DevTmp t_rdbg;
        if ((kn.timing & TIMING_WG) && !h->async_loop) {
            HIP_TRY(h, hipMalloc(&t_rdbg.p, sizeof(unsigned long long) * 16 * (size_t)nblk_dbg));
            HIP_TRY(h, hipMemsetAsync(t_rdbg.p, 0, sizeof(unsigned long long) * 16 * (size_t)nblk_dbg, stream));
            P.DBG = t_rdbg.as<unsigned long long>();
        }
        prof.next(5, stream);
        launch_wg(hyb_bx, nullptr, ctl + PIPE_WG, h->async_loop ? nullptr : ctl + PIPE_ABORT);
        prof.end(stream);
        h->last_mode = 2;
        if (h->async_loop) {             // closed-loop driver: nothing comes back to the host per step (this path has no abort word)
            HIP_TRY(h, hipGetLastError());
            h->async_ok = true;
            return MPC_OK;
        }
        HIP_TRY(h, wait_stream(h, stream));          // (the last workgroup to leave has put the block's head into h_pipe)
        h->h_fail[0] = h->h_pipe[14];
        for (int q = 0; q < 4; ++q) h->h_fail[2 + q] = h->h_pipe[16 + q];
        if (h->resc_ran && !h->in_rescue) h->rescued_last = (int)h->h_pipe[20];
        report_wtrace();
        if (P.DBG) {          // shader-clock stamps of every workgroup's third round
            std::vector<unsigned long long> hd((size_t)16 * nblk_dbg);
            HIP_TRY(h, hipMemcpy(hd.data(), P.DBG, hd.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double acc[16] = {0};
            int cnt = 0;
            {
                // k_solve_wg: 12 round start, 13 records in LDS, 14 sweeps done, [0..10 stage_block's own stamps], 15 round end
                const int order[16] = {12, 13, 14, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15, 15};
                for (int bq = 0; bq < nblk_dbg; ++bq) {
                    const unsigned long long* r = hd.data() + (size_t)bq * 16;
                    if (!r[15] || !r[10]) continue;
                    for (int q = 0; q < 14; ++q) acc[q] += (double)(long long)(r[order[q + 1]] - r[order[q]]);
                    ++cnt;
                }
                static const char* names[14] = {"records", "sweeps", "enter", "load+premath", "or", "P1", "reduce1+ls-begin", "linesearch", "P3-update", "exchange", "P4-eval", "reduce3", "P5", "drain"};
                fprintf(stderr, "[mpcgpu k_solve_wg timing, shader-clock ticks, third round of %d workgroups]", cnt);
                for (int q = 0; q < 14; ++q) fprintf(stderr, " %s=%.0f", names[q], cnt ? acc[q] / cnt : 0.0);
            }
            fprintf(stderr, "\n");
            P.DBG = nullptr;
        }
        piped = true;
        it = (int)h->h_fail[2];
        h->res_prof[1] = 1; h->res_prof[2] = it; h->res_prof[3] = nblk_dbg; h->res_prof[4] = h->h_fail[3]; h->res_prof[5] = h->h_fail[4]; h->res_prof[6] = h->h_fail[5];
    } else {
        // (measured: 7-11 % faster than one launch per kernel at B = 64 ... 1024, 31 % at B = 4096; at B = 8192 the early
        //  finishers of converged mode still gain 15 %, a fixed iteration count loses 8 % -- with two tiles per Riccati
        //  worker both roles are throughput bound and the split of the CUs only costs)
        if (pipe_path) {
            PipeArgs A;
            A.ntiles = (uint32_t)ntiles;
            A.n_ric = (uint32_t)n_ric;
            A.xcd_mask = xcd_mask;
            A.items = 64u / (uint32_t)bx;
            A.cap = pipe_cap;
            A.flags = kn.pipe_test_abort ? 2u : 0u;
            // the Riccati workers help with the stage items (k_pipeline<.., HELP>) where those are the bottleneck: more than three per stage worker and round
            const bool pipe_help = (kn.pipe_help < 0 ? (int)A.items * tiles_x > 3 * (cu_x - n_ric) : kn.pipe_help != 0);
            A.handover = (uint32_t)hand;
            int32_t* ho_list = nullptr;
            A.ho_list = nullptr;
            if (hand > 0) {
                ho_list = static_cast<int32_t*>(scratch_get(h, 36, (size_t)HO_BUCKETS * Bp * sizeof(int32_t)));
                if (!ho_list) { h->err = "out of device memory"; return MPC_ERR_HIP; }
                A.ho_list = ho_list;
            }
            A.ctl = ctl;
            unsigned long long* d_pdbg = nullptr;
            if (kn.timing & TIMING_PIPE) {
                HIP_TRY(h, hipMalloc(&t_pdbg.p, sizeof(unsigned long long) * 16 * (size_t)h->n_cu));
                d_pdbg = t_pdbg.as<unsigned long long>();
                HIP_TRY(h, hipMemsetAsync(d_pdbg, 0, sizeof(unsigned long long) * 16 * (size_t)h->n_cu, stream));
                P.DBG = d_pdbg;
            }
            prof.next(3, stream);
            // tiles about to leave the pipeline (at most hand + 16 instances left; never a full tile) write the mailbox arrays in their stage items
            P.mbw_live = hand > 0 ? std::min(63, hand + 16) : 0;
            // (helping Riccati workers only in the variant with the reference's bound structure compiled in: the general one is at 512 registers with them)
            if (pipe_help && masked) hipLaunchKernelGGL((k_pipeline<NX, 2, true>), dim3(h->n_cu), dim3(threads), std::max(lds_bytes, ric_lds), stream, P, A, h->hp.n_mult, h->hp.n_z, stash_rows);
            else if (masked) hipLaunchKernelGGL((k_pipeline<NX, 2>), dim3(h->n_cu), dim3(threads), std::max(lds_bytes, ric_lds), stream, P, A, h->hp.n_mult, h->hp.n_z, stash_rows);
            else hipLaunchKernelGGL((k_pipeline<NX, false>), dim3(h->n_cu), dim3(threads), std::max(lds_bytes, ric_lds), stream, P, A, h->hp.n_mult, h->hp.n_z, stash_rows);
            if (hand <= 0) prof.end(stream);
            if (hand > 0) {        // (its statistics words are part of the control block: no fill, no copy of their own)
                if (!h->prof_span) prof.next(5, stream);          // (span mode: the pipeline's span stays open over k_solve_wg)
                if (ho_list) {
                    // (the counters: words of the control block, zero at the start of every solve)
                    wg_list = ho_list; wg_list_n = ctl + PIPE_HO;
                    // as many workgroups as the machine holds at once (four single-wavefront workgroups per CU), each with up to hyb_bx instances
                    wg_grid = std::min((B + hyb_bx - 1) / hyb_bx, std::max(4 * h->n_cu, (int)((size_t)hand * ntiles + hyb_bx - 1) / hyb_bx));
                }
                // (the last kernel of the solve: its last workgroup copies the head of the control block into the pinned host block)
                launch_wg(hyb_bx, (const uint32_t*)(ctl + PIPE_ABORT), ctl + PIPE_WG, h->async_loop ? nullptr : ctl + PIPE_ABORT);
            }
            prof.end(stream);
            if (h->async_loop) {
                // closed-loop driver: nothing comes back to the host per step -- a launch that had to be abandoned leaves its mark
                // in the loop's sticky abort word, the bookkeeping kernels behind it then do nothing and the host replays the loop
                hipLaunchKernelGGL(k_loop_sticky, dim3(1), dim3(1), 0, stream, (const uint32_t*)(ctl + PIPE_ABORT), h->d_fail + 1);
                HIP_TRY(h, hipGetLastError());
                h->async_ok = true;
                h->last_mode = 1;
                return MPC_OK;
            }
            // (no kernel behind the loop: the rows are out -- Params::emit -- and k_solve_wg has left the block's head in h_pipe; a pipeline that
            //  runs its tiles to the end, option hybrid = 0, has no such epilogue: a copy command)
            if (hand <= 0) HIP_TRY(h, hipMemcpyAsync(h->h_pipe, ctl + PIPE_ABORT, PIPE_FIN_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(h, wait_stream(h, stream));
            h->h_fail[0] = h->h_pipe[14];
            report_wtrace();
            for (int q = 0; q < 4; ++q) h->h_fail[2 + q] = h->h_pipe[16 + q];
            if (h->resc_ran && !h->in_rescue) h->rescued_last = (int)h->h_pipe[20];       // (fifth word: instances that took the second chance inside the launch)
            if (hand > 0) {        // the stragglers' kernel: rounds of its slowest workgroup, workgroups, workgroup-rounds, sweeps, instance-iterations
                h->res_prof[1] = 1; h->res_prof[2] = h->h_fail[2]; h->res_prof[3] = wg_grid > 0 ? wg_grid : (B + hyb_bx - 1) / hyb_bx;      // (workgroups launched: what the machine holds when they are dealt from the hand-over lists)
                h->res_prof[4] = h->h_fail[3]; h->res_prof[5] = h->h_fail[4]; h->res_prof[6] = h->h_fail[5];
            }
            if (h->h_pipe[0] != 0u) {
                // a bounded wait ran out (e.g. the dispatcher left an XCD without stage workers): the workspace is part-way
                // through an iteration, so start over with one launch per kernel.  Once may be a transient of the machine's other tenants (the
                // workgroups of a collective spinning on a slow peer, another handle's persistent launch); the handle stays on that path when it
                // has happened PIPE_ABORTS_MAX times
                h->pipe_disabled = true;
                h->resc_in_kernel = h->resc_ran = false;         // (the k_solve_wg behind the abandoned launch returned at once: no instance has had its second chance)
                // (abort word: 1 option pipe_test_abort, 2 a Riccati worker waited for its tile's stage items [tile << 16 | round; arrivals], 3 a helping
                //  Riccati worker / 4 a stage worker waited for its queue slot [ticket; queue tail]; words 28 - 31 of the block's head)
                fprintf(stderr, "[mpcgpu] single-launch pipeline abandoned (bounded wait expired: code %u, %u / %u, a millisecond later %u, waited %u ticks); re-running with one launch per kernel\n",
                        h->h_pipe[0], h->h_pipe[28], h->h_pipe[29], h->h_pipe[30], h->h_pipe[31]);
                {   // (the arrival counters of all tiles as the launch left them: a tile short of a multiple of its items lost an arrival, one above received a foreign one)
                    std::vector<uint32_t> sd((size_t)ntiles);
                    if (hipMemcpy(sd.data(), ctl + PIPE_HDR, sd.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess) {
                        fprintf(stderr, "[mpcgpu]   arrivals per tile (items per tile and round: %u):", A.items);
                        for (int q = 0; q < ntiles; ++q) fprintf(stderr, " %u%s", sd[q] & 0x7FFFFFFFu, (sd[q] >> 31) ? "r" : "");
                        fprintf(stderr, "\n");
                    }
                }
                const int rc_again = solve_dev_impl<NX>(h, B, d_x0, d_p, d_obst, d_x_out, d_status, d_iters, d_kkt, stream, trace, trace_rows, n_it_out);
                if (++h->pipe_aborts < mpc_handle::PIPE_ABORTS_MAX) h->pipe_disabled = false;
                return rc_again;
            }
            P.DBG = nullptr;
            if (d_pdbg) {       // shader-clock stamps of every worker's LAST work item / tile pass
                std::vector<unsigned long long> hd((size_t)16 * h->n_cu);
                HIP_TRY(h, hipMemcpy(hd.data(), d_pdbg, hd.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                double sa[16] = {0}, ra[9] = {0};
                int ns = 0, nr = 0;
                for (int bq = 0; bq < h->n_cu; ++bq) {
                    const unsigned long long* r = hd.data() + (size_t)bq * 16;
                    if (r[15] && r[10]) {
                        const int order[16] = {11, 12, 13, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 15};
                        for (int q = 0; q + 1 < 16; ++q) sa[q] += (double)(long long)(r[order[q + 1]] - r[order[q]]);
                        ++ns;
                    } else if (r[13] && r[2]) {
                        ra[0] += (double)(long long)(r[12] - r[11]); ra[1] += (double)(long long)(r[1] - r[12]);
                        ra[2] += (double)(long long)(r[2] - r[1]); ra[3] += (double)(long long)(r[13] - r[2]);
                        ra[4] += (double)(long long)(r[4] - r[3]); ra[5] += (double)(long long)(r[5] - r[4]);
                        ra[6] += (double)(long long)(r[7] - r[6]); ra[7] += (double)(long long)(r[8] - r[7]);
                        ra[8] += (double)(long long)(r[9] - r[1]);
                        ++nr;
                    }
                }
                static const char* sn[15] = {"dequeue", "acquire+bcast", "enter", "issue-loads", "wait+barrier", "P1", "reduce1", "linesearch", "P3-update",
                                             "exchange", "P4-eval", "reduce3", "P5", "drain", "signal"};
                fprintf(stderr, "[mpcgpu pipeline timing, shader-clock ticks, last item of %d stage workers]", ns);
                for (int q = 0; q < 15; ++q) fprintf(stderr, " %s=%.0f", sn[q], ns ? sa[q] / ns : 0.0);
                fprintf(stderr, "\n[last pass of %d Riccati workers] wait=%.0f backward=%.0f forward=%.0f publish=%.0f; stage 15 of the backward sweep: barrier=%.0f step=%.0f, of the forward sweep: barrier=%.0f step=%.0f, its first stage starts %.0f ticks after the backward sweep\n",
                        nr, nr ? ra[0] / nr : 0.0, nr ? ra[1] / nr : 0.0, nr ? ra[2] / nr : 0.0, nr ? ra[3] / nr : 0.0, nr ? ra[4] / nr : 0.0, nr ? ra[5] / nr : 0.0,
                        nr ? ra[6] / nr : 0.0, nr ? ra[7] / nr : 0.0, nr ? ra[8] / nr : 0.0);
            }
            piped = true;
            h->last_mode = 1;
            it = (int)h->h_pipe[1];
            const unsigned long long* st64 = reinterpret_cast<const unsigned long long*>(h->h_pipe + 2);
            h->pipe_prof[1] = 1; h->pipe_prof[2] = it;
            h->pipe_prof[3] = (double)st64[0] * 1e-5; h->pipe_prof[4] = (double)st64[1] * 1e-5; h->pipe_prof[5] = (double)st64[2] * 1e-5;   // 100 MHz ticks -> ms
            h->pipe_prof[6] = (double)st64[3]; h->pipe_prof[7] = (double)h->h_pipe[12] + 1e-3 * (double)h->h_pipe[13];
        }
    }
    if (!piped) {
        // one launch per kernel: the poll table of this solve and (unless an asynchronous closed loop accumulates over its steps) the
        // count of instances that do not converge start from zero -- the pipeline path keeps both in its control block
        if (!h->async_loop) HIP_TRY(h, hipMemsetAsync(h->d_fail, 0, sizeof(uint32_t), stream));
        if (polled) HIP_TRY(h, hipMemsetAsync(h->d_counter, 0, sizeof(int32_t) * mpc_handle::MAX_GROUPS * mpc_handle::MAX_POLL_IT, stream));
        if (G > 1) {     // the sub-streams were forked before these memsets were enqueued
            HIP_TRY(h, hipEventRecord(h->ev_fork, stream));
            for (int g = 0; g < G; ++g) HIP_TRY(h, hipStreamWaitEvent(h->sub_stream[g], h->ev_fork, 0));
        }
    }
    while (!piped && it < cap) {
        const int n = std::min(trace ? 1 : chunk, cap - it);
        bool any = false;
        for (int j = 0; j < n; ++j) {
            for (int g = 0; g < G; ++g) {                 // interleave the groups launch by launch
                const Group& q = grp[g];
                if (!q.running) continue;
                any = true;
                Params Pg = P;
                Pg.tile0 = q.tile0;
                prof.begin(0, q.st);
                if (stage_timing && it + j == 3) Pg.DBG = d_dbg + (size_t)8 * nblk;   // Riccati stamps of the 4th iteration
                hipLaunchKernelGGL((k_riccati<NX>), dim3(q.ntl), dim3(192), ric_lds, q.st, Pg);
                Pg.DBG = nullptr;
                prof.end(q.st);
                prof.begin(1, q.st);
                P.DBG = (stage_timing && it + j == 2) ? d_dbg : nullptr;      // stamp the third iteration
                P.run_counter = polled ? h->d_counter + (size_t)g * mpc_handle::MAX_POLL_IT + (it + j) : nullptr;
                launch_stage(q, false);
                P.run_counter = nullptr;
                prof.end(q.st);
            }
            if (trace) { rc = record_trace(it + j); if (rc) return rc; }
        }
        it += n;
        if (!any) break;
        if (!polled) {
            if (trace) {                                  // trace mode: exact stop, one synchronous count per iteration
                HIP_TRY(h, hipMemsetAsync(h->d_counter, 0, sizeof(int32_t), stream));
                hipLaunchKernelGGL(k_count_running, dim3((B + 255) / 256), dim3(256), 0, stream, h->d_iws, (uint32_t)w.itile_elems, 0, B, h->d_counter);
                HIP_TRY(h, hipMemcpyAsync(h->h_counter, h->d_counter, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                HIP_TRY(h, hipStreamSynchronize(stream));
                if (h->h_counter[0] == 0) break;
            }
            continue;
        }
        const int slot = chunk_id & 1;
        for (int g = 0; g < G; ++g) {
            const Group& q = grp[g];
            if (!q.running) continue;
            HIP_TRY(h, hipMemcpyAsync(h->h_counter + 2 * g + slot, h->d_counter + (size_t)g * mpc_handle::MAX_POLL_IT + (it - 1), sizeof(int32_t),
                                      hipMemcpyDeviceToHost, q.st));
            HIP_TRY(h, hipEventRecord(h->ev_poll[g][slot], q.st));
        }
        if (chunk_id >= 1) {                              // look at the PREVIOUS chunk's poll (this chunk is already in the queue)
            bool still = false;
            for (int g = 0; g < G; ++g) {
                Group& q = grp[g];
                if (!q.running) continue;
                HIP_TRY(h, hipEventSynchronize(h->ev_poll[g][slot ^ 1]));
                if (h->h_counter[2 * g + (slot ^ 1)] == 0) q.running = false; else still = true;
            }
            if (!still) break;
        }
        ++chunk_id;
    }
    P.DBG = nullptr;
    if (trace) { rc = record_trace(it); if (rc) return rc; }
    if (stage_timing) {
        HIP_TRY(h, hipStreamSynchronize(stream));
        std::vector<unsigned long long> hd((size_t)16 * nblk);
        HIP_TRY(h, hipMemcpy(hd.data(), d_dbg, hd.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double acc[10] = {0};
        int cnt = 0;
        for (int bq = 0; bq < nblk; ++bq) {
            if (!hd[(size_t)bq * 16 + 10]) continue;
            for (int q = 0; q < 10; ++q) acc[q] += (double)(hd[(size_t)bq * 16 + q + 1] - hd[(size_t)bq * 16 + q]);
            ++cnt;
        }
        fprintf(stderr, "[mpcgpu stage timing, shader-clock ticks per block, mean over %d blocks]", cnt);
        static const char* names[10] = {"issue-loads", "wait+barrier", "P1", "reduce1", "linesearch", "P3-update", "exchange", "P4-eval", "reduce3", "P5"};
        for (int q = 0; q < 10; ++q) fprintf(stderr, " %s=%.0f", names[q], cnt ? acc[q] / cnt : 0.0);
        fprintf(stderr, "\n");
        {
            double bw = 0, fw = 0;
            int c2 = 0;
            for (int tq = 0; tq < ntiles && (size_t)(8 * nblk + (tq + 1) * 16) <= hd.size(); ++tq) {
                const unsigned long long* r = hd.data() + (size_t)8 * nblk + (size_t)tq * 16;
                if (!r[2]) continue;
                bw += (double)(r[1] - r[0]);
                fw += (double)(r[2] - r[1]);
                ++c2;
                if (tq == 0) fprintf(stderr, "[riccati stage 15 of tile 0] bwd barrier-wait=%lld compute=%lld | fwd barrier-wait=%lld compute=%lld\n",
                                     (long long)(r[4] - r[3]), (long long)(r[5] - r[4]), (long long)(r[7] - r[6]), (long long)(r[8] - r[7]));
            }
            fprintf(stderr, "[mpcgpu riccati timing, ticks per workgroup, mean over %d] backward=%.0f forward=%.0f\n", c2, c2 ? bw / c2 : 0.0, c2 ? fw / c2 : 0.0);
        }
    }
    if (n_it_out) *n_it_out = it;
    for (int g = 0; g < G; ++g) {
        const Group& q = grp[g];
        if (q.b1 <= q.b0 || piped) continue;          // (pipeline: already enqueued behind the launch)
        Params Pg = P;
        Pg.tile0 = q.tile0;
        prof.begin(2, q.st);
        hipLaunchKernelGGL((k_egest<NX>), dim3(q.ntl, (2 * d.N + NX * (d.N + 1) + 63) / 64), dim3(256), 0, q.st, Pg, (const uint32_t*)nullptr, h->d_fail, (uint32_t*)nullptr, 0u);
        prof.end(q.st);
        if (G > 1) {
            HIP_TRY(h, hipEventRecord(h->ev_join[g], q.st));
            HIP_TRY(h, hipStreamWaitEvent(stream, h->ev_join[g], 0));
        }
    }
    HIP_TRY(h, hipGetLastError());
    h->prof[5] = it;
    if (!piped) HIP_TRY(h, hipMemcpyAsync(h->h_fail, h->d_fail, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (d.fixed_iters <= 0 || h->profiling) HIP_TRY(h, hipStreamSynchronize(stream));
    const double its = it;
    prof.collect();
    h->prof[5] = its;
    return MPC_OK;
}

static int solve_dev_any(mpc_handle* h, int32_t B, const double* d_x0, const double* d_p, const double* d_obst, double* d_x_out,
                         int32_t* d_status, int32_t* d_iters, double* d_kkt, hipStream_t stream, double* trace, int32_t trace_rows, int32_t* n_it) {
    if (h->hp.desc.nx == 5)
        return solve_dev_impl<5>(h, B, d_x0, d_p, d_obst, d_x_out, d_status, d_iters, d_kkt, stream, trace, trace_rows, n_it);
    return solve_dev_impl<6>(h, B, d_x0, d_p, d_obst, d_x_out, d_status, d_iters, d_kkt, stream, trace, trace_rows, n_it);
}

// Second chance for the instances of a batch that did not converge, on the device and behind the C-ABI.  IPOPT hands a start
// that is locally infeasible -- typically a guess that runs straight through the obstacle, where the linearised circle rows
// cannot be met within the fraction-to-the-boundary rule -- to its feasibility-restoration phase, which is not restated here
// (DESIGN.md section 2).  Instead: the failed instances are compacted into a sub-batch and re-solved with the lower bound of the
// circle-distance rows (optimizer.py:426-428) raised in steps from 0 to its true value, each level warm-started from the last
// one that converged; the last level is the ORIGINAL problem, so what is written back is a KKT point of the original NLP to the
// original tolerance -- or nothing (the original failure stays).  Two passes: levels {0, 1}, then {0, 0.4, 0.7, 0.9, 1} for
// what is still open.  Everything stays in device memory; the only host traffic is the count of open instances.
static int rescue_dev(mpc_handle* h, int32_t B, const double* d_x0, const double* d_p, const double* d_obst, double* d_x_out,
                      int32_t* d_status, int32_t* d_iters, double* d_kkt, hipStream_t stream) {
    const size_t nw = h->hp.n_w(), nB = (size_t)B;
    int32_t* idx = static_cast<int32_t*>(scratch_get(h, 14, (nB + 1) * 4));
    if (!idx) { h->err = "rescue: out of device memory"; return MPC_ERR_HIP; }
    int32_t* cnt = idx + nB;
    static const double pass1[] = {0.0, 1.0}, pass2[] = {0.0, 0.4, 0.7, 0.9, 1.0};
    const double ol_keep = h->hp.ol, tol_keep = h->hp.desc.tol;
    h->rescued_last = 0;
    int rc = MPC_OK;
    for (int pass = 0; pass < 2 && rc == MPC_OK; ++pass) {
        hipLaunchKernelGGL(k_rescue_select, dim3(1), dim3(1024), 0, stream, d_status, B, idx, cnt);
        int32_t n = 0;
        HIP_TRY(h, hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(h, hipStreamSynchronize(stream));
        if (n <= 0) break;
        // sub-batch buffers: [xs | ps | out] rows, obstacle rows, status / iterations / kkt of a level, accumulated iterations
        double* buf = static_cast<double*>(scratch_get(h, 15, ((size_t)n * (3 * nw + 6 + 1)) * 8 + (size_t)n * 3 * 4 + 64));
        if (!buf) { h->err = "rescue: out of device memory"; return MPC_ERR_HIP; }
        double *xs = buf, *ps = xs + (size_t)n * nw, *out = ps + (size_t)n * nw, *os = out + (size_t)n * nw, *kk = os + (size_t)n * 6;
        int32_t *st = reinterpret_cast<int32_t*>(kk + n), *it = st + n, *acc = it + n;
        hipLaunchKernelGGL(k_rescue_gather, dim3(n), dim3(128), 0, stream, idx, (int)nw, d_x0, d_p, d_obst, xs, ps, os, acc);
        const double* fr = pass == 0 ? pass1 : pass2;
        const int nfr = pass == 0 ? 2 : 5;
        for (int q = 0; q < nfr && rc == MPC_OK; ++q) {
            h->hp.ol = relax_lo(fr[q] * h->hp.ol_raw);
            h->hp.desc.tol = (q + 1 < nfr) ? std::max(tol_keep, 1e-4) : tol_keep;      // intermediate levels only produce warm starts
            h->in_rescue = true;
            rc = solve_dev_any(h, n, xs, ps, d_obst ? os : nullptr, out, st, it, kk, stream, nullptr, 0, nullptr);
            h->in_rescue = false;
            if (rc == MPC_OK) hipLaunchKernelGGL(k_rescue_carry, dim3(n), dim3(128), 0, stream, (int)nw, st, it, out, xs, acc);
        }
        h->hp.ol = ol_keep;
        h->hp.desc.tol = tol_keep;
        if (rc == MPC_OK) hipLaunchKernelGGL(k_rescue_scatter, dim3(n), dim3(128), 0, stream, idx, (int)nw, st, out, kk, acc, d_x_out, d_status, d_iters, d_kkt);
        if (pass == 0) h->rescued_last = n;
    }
    h->hp.ol = ol_keep;
    h->hp.desc.tol = tol_keep;
    if (rc == MPC_OK) { HIP_TRY(h, hipGetLastError()); HIP_TRY(h, hipStreamSynchronize(stream)); }
    return rc;
}

// instances one solve can take: the workspace (tile-major section + mailbox arrays, both linear in the number of tiles) is addressed
// with 32-bit buffer offsets
// can a batch of this handle run in the persistent launch at all -- the conditions of solve_dev_impl that do not depend on the batch size
static bool pipeline_shape(const mpc_handle* h) {
    const mpc_problem_desc& d = h->hp.desc;
    const mpc_handle::Knobs& kn = h->knobs;
    if (4 * (d.N + 1) > 256 || kn.big_wg || !kn.pipeline || h->pipe_disabled || kn.groups > 0 || (kn.timing & TIMING_STAGE) || d.fixed_iters > 0) return false;
    const int threads = (((d.N + 1) * pick_bx(d.N, 256) + 63) / 64) * 64;
    return threads >= 192 && threads <= 256;
}
static size_t max_rows_per_solve(const mpc_handle* h) {
    const WsLayout w1 = ws_layout(h->hp.desc.N, h->hp.desc.nx, 64, wants_mailbox(h));
    size_t max_b = (((size_t)1 << 32) - 1) / (w1.total * sizeof(double)) * 64;
    // (the persistent launch takes PIPE_MAX_TILES tiles -- 128 until the end of round 6, when a batch beyond them fell back to one launch per kernel
    //  and iteration: 43 % more time per instance at B = 8193 than at 8192; now 256, four tiles per Riccati worker, and chunks of that size beyond:
    //  B = 8193 2.33 -> 1.37 ms, 12288 3.06 -> 1.96 ms, 16384 2.63 ms; N = 10: 16384 1.0 ms = 16 M instances/s)
    if (pipeline_shape(h)) max_b = std::min(max_b, (size_t)PIPE_MAX_TILES * 64);
    if (h->knobs.max_batch > 0) max_b = std::min(max_b, (size_t)(h->knobs.max_batch + 63) / 64 * 64);
    return max_b;
}

static int solve_dev(mpc_handle* h, int32_t B, const double* d_x0, const double* d_p, const double* d_obst, double* d_x_out,
                     int32_t* d_status, int32_t* d_iters, double* d_kkt, hipStream_t stream, double* trace, int32_t trace_rows,
                     int32_t* n_it) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || !d_x0 || !d_p || !d_x_out) { h->err = "B > 0 and x0, p, x_out are required"; return MPC_ERR_INVALID; }
    if (!h->hp.bounds_set) { h->err = "mpc_set_bounds has not been called"; return MPC_ERR_STATE; }
    HIP_TRY(h, hipSetDevice(h->device));
    h->rescued_last = 0;
    // the second chance needs the per-instance status: an internal row when the caller did not ask for it
    const bool rescue = h->knobs.rescue && h->hp.desc.fixed_iters <= 0 && !trace && h->hp.has_ol && h->hp.ol_raw > 0.0;
    if (rescue && !d_status) {
        d_status = static_cast<int32_t*>(scratch_get(h, 11, (size_t)B * 4));
        if (!d_status) { h->err = "out of device memory"; return MPC_ERR_HIP; }
    }
    // The workspace is addressed with 32-bit buffer offsets (< 4 GiB): a batch beyond that is solved in chunks of whole tiles, one after
    // the other on the same stream (instances are independent; the rows of a chunk are a contiguous slice of every caller buffer).
    {
        const size_t max_b = max_rows_per_solve(h);
        if ((size_t)B > max_b && !trace && !h->async_loop) {
            const size_t nw = h->hp.n_w();
            int rescued = 0;
            for (size_t off = 0; off < (size_t)B; off += max_b) {
                const int32_t n = (int32_t)std::min(max_b, (size_t)B - off);
                const int rcc = solve_dev(h, n, d_x0 + off * nw, d_p + off * nw, d_obst ? d_obst + off * 6 : nullptr, d_x_out + off * nw,
                                          d_status ? d_status + off : nullptr, d_iters ? d_iters + off : nullptr, d_kkt ? d_kkt + off : nullptr, stream,
                                          nullptr, 0, nullptr);
                if (rcc) return rcc;
                rescued += h->rescued_last;
            }
            h->rescued_last = rescued;
            return MPC_OK;
        }
    }
    h->resc_in_kernel = h->resc_ran = false;
    const int rc = solve_dev_any(h, B, d_x0, d_p, d_obst, d_x_out, d_status, d_iters, d_kkt, stream, trace, trace_rows, n_it);
    // (converged mode: the solve has synchronised the stream; a launch of k_solve_wg with the second chance inside has given every stalled
    //  instance its levels already)
    // (what the next solve of this handle does about stalled instances: see resc_cond)
    if (rc == MPC_OK && rescue && !h->async_ok) h->resc_hint = h->h_fail[0] != 0u || h->rescued_last > 0;
    if (rc != MPC_OK || !rescue || h->h_fail[0] == 0u || h->resc_in_kernel) return rc;
    double prof_keep[6], pipe_keep[8];
    const int mode_keep = h->last_mode;
    memcpy(prof_keep, h->prof, sizeof prof_keep);
    memcpy(pipe_keep, h->pipe_prof, sizeof pipe_keep);
    const int in_kernel = h->rescued_last;                  // (instances that had their second chance inside k_solve_wg already)
    const int rr = rescue_dev(h, B, d_x0, d_p, d_obst, d_x_out, d_status, d_iters, d_kkt, stream);
    h->rescued_last += in_kernel;
    memcpy(h->prof, prof_keep, sizeof prof_keep);          // the measurement helpers describe the main solve
    memcpy(h->pipe_prof, pipe_keep, sizeof pipe_keep);
    h->last_mode = mode_keep;
    return rr;
}

static int ensure_io(mpc_handle* h, size_t B) {
    if (B <= h->cap_io) return MPC_OK;
    free_io(h);
    const size_t nw = h->hp.n_w();
    HIP_TRY(h, hipMalloc(&h->d_x0, B * nw * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->d_p, B * nw * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->d_xout, B * nw * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->d_kkt, B * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->d_obst, B * 6 * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->d_status, B * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc(&h->d_iters, B * sizeof(int32_t)));
    h->cap_io = B;
    return MPC_OK;
}

static int solve_host(mpc_handle* h, int32_t B, const double* x0, const double* p, const double* obst, double* x_out,
                      int32_t* status, int32_t* iters, double* kkt, double* trace, int32_t trace_rows, int32_t* n_it) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || !x0 || !p || !x_out) { h->err = "B > 0 and x0, p, x_out are required"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_io(h, (size_t)B);
    if (rc) return rc;
    const size_t nw = h->hp.n_w();
    hipStream_t s = h->own_stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_x0, x0, B * nw * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_p, p, B * nw * sizeof(double), hipMemcpyHostToDevice, s));
    if (obst) HIP_TRY(h, hipMemcpyAsync(h->d_obst, obst, (size_t)B * 6 * sizeof(double), hipMemcpyHostToDevice, s));
    rc = solve_dev(h, B, h->d_x0, h->d_p, obst ? h->d_obst : nullptr, h->d_xout, h->d_status, h->d_iters, h->d_kkt, s, trace,
                   trace_rows, n_it);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(x_out, h->d_xout, B * nw * sizeof(double), hipMemcpyDeviceToHost, s));
    if (status) HIP_TRY(h, hipMemcpyAsync(status, h->d_status, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (iters) HIP_TRY(h, hipMemcpyAsync(iters, h->d_iters, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (kkt) HIP_TRY(h, hipMemcpyAsync(kkt, h->d_kkt, B * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return MPC_OK;
}

extern "C" {

int mpc_solve_batch_dev(mpc_handle* h, int32_t B, const double* d_x0, const double* d_p, const double* d_obst, double* d_x_out,
                        int32_t* d_status, int32_t* d_iters, double* d_kkt, void* stream) {
    return solve_dev(h, B, d_x0, d_p, d_obst, d_x_out, d_status, d_iters, d_kkt, (hipStream_t)stream, nullptr, 0, nullptr);
}

int mpc_solve_batch(mpc_handle* h, int32_t B, const double* x0, const double* p, const double* obst, double* x_out,
                    int32_t* status, int32_t* iters, double* kkt) {
    return solve_host(h, B, x0, p, obst, x_out, status, iters, kkt, nullptr, 0, nullptr);
}

int mpc_solve_batch_trace(mpc_handle* h, int32_t B, const double* x0, const double* p, const double* obst, double* x_out,
                          int32_t* status, int32_t* iters, double* kkt, double* trace, int32_t trace_rows, int32_t* n_it) {
    return solve_host(h, B, x0, p, obst, x_out, status, iters, kkt, trace, trace_rows, n_it);
}

int mpc_plant_step_dev(mpc_handle* h, int32_t B, int32_t integrator, const double* d_x, const double* d_u, double* d_x_next, void* stream_) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || !d_x || !d_u || !d_x_next || integrator < 0 || integrator > 1) { h->err = "bad plant-step arguments"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    const int nx = h->hp.desc.nx;
    hipStream_t s = (hipStream_t)stream_;
    Params P{};
    P.dt = h->hp.desc.dt; P.wheelbase = h->hp.desc.wheelbase; P.nx = nx;
    if (nx == 5) hipLaunchKernelGGL((k_plant_step<5>), dim3((B + 255) / 256), dim3(256), 0, s, P, d_x, d_u, d_x_next, B, integrator);
    else hipLaunchKernelGGL((k_plant_step<6>), dim3((B + 255) / 256), dim3(256), 0, s, P, d_x, d_u, d_x_next, B, integrator);
    HIP_TRY(h, hipGetLastError());
    return MPC_OK;
}

int mpc_plant_step(mpc_handle* h, int32_t B, int32_t integrator, const double* x, const double* u, double* x_next) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || !x || !u || !x_next || integrator < 0 || integrator > 1) { h->err = "bad plant-step arguments"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nx = (size_t)h->hp.desc.nx, nB = (size_t)B;
    double* dx = static_cast<double*>(scratch_get(h, 0, nB * nx * 8));       // buffers stay with the handle: optimize() calls this every step
    double* du = static_cast<double*>(scratch_get(h, 1, nB * 2 * 8));
    double* dn = static_cast<double*>(scratch_get(h, 2, nB * nx * 8));
    if (!dx || !du || !dn) { h->err = "plant step: out of device memory"; return MPC_ERR_HIP; }
    hipStream_t s = h->own_stream;
    HIP_TRY(h, hipMemcpyAsync(dx, x, nB * nx * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(du, u, nB * 2 * 8, hipMemcpyHostToDevice, s));
    const int rc = mpc_plant_step_dev(h, B, integrator, dx, du, dn, (void*)s);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(x_next, dn, nB * nx * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return MPC_OK;
}

int mpc_closed_loop_batch_dev_ex(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* d_init_state, const double* d_path,
                                 const double* d_orient, const double* d_vdes, int32_t noise_mode, double sigma, uint64_t seed,
                                 double* d_traj, double* d_ctrl, int32_t* d_step_status, void* stream_) {
    if (!h) return MPC_ERR_INVALID;
    const mpc_problem_desc& d = h->hp.desc;
    if (B <= 0 || L <= 0 || Lp < L || L < d.N || !d_init_state || !d_path || !d_orient || !d_vdes || !d_traj || !d_ctrl) {
        h->err = "closed loop: B > 0, L >= N, Lp >= L and all of init_state, path, orient, vdes, traj, ctrl are required";
        return MPC_ERR_INVALID;
    }
    if (noise_mode < 0 || noise_mode > 2 || (noise_mode != 0 && !(sigma >= 0.0))) { h->err = "closed loop: noise_mode in {0, 1, 2}, sigma >= 0"; return MPC_ERR_INVALID; }
    if (!h->hp.bounds_set) { h->err = "mpc_set_bounds has not been called"; return MPC_ERR_STATE; }
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_io(h, (size_t)B);
    if (rc) return rc;
    if (h->cap_state < (size_t)B) {
        if (h->d_state) (void)hipFree(h->d_state);
        h->d_state = nullptr;
        HIP_TRY(h, hipMalloc(&h->d_state, (size_t)B * 6 * sizeof(double)));
        h->cap_state = (size_t)B;
    }
    hipStream_t stream = (hipStream_t)stream_;
    LoopArgs A{};
    A.B = B; A.N = d.N; A.L = L; A.Lp = Lp; A.nx = d.nx;
    A.init_state = d_init_state; A.path = d_path; A.orient = d_orient; A.vdes = d_vdes;
    A.state = h->d_state; A.x0 = h->d_x0; A.p = h->d_p; A.x_out = h->d_xout; A.status = h->d_status;
    A.traj = d_traj; A.ctrl = d_ctrl; A.step_status = d_step_status;
    A.noise_mode = noise_mode; A.sigma = sigma; A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32);
    Params P{};
    P.dt = d.dt; P.wheelbase = d.wheelbase; P.nx = d.nx;
    const dim3 grid((B + 127) / 128), block(128);
    // First attempt: the whole loop enqueued without a single host synchronisation (every solve in the persistent pipeline
    // launch, which needs no convergence poll).  What could go wrong on the way -- a pipeline launch abandoned, an instance
    // that needs the second chance -- is recorded on the device and looked at ONCE, at the end; then the loop is replayed
    // step by step with the host in between (the per-kernel path polls for convergence, the second chance needs the count).
    bool replay = true, loop_abandoned = false;
    // (a batch beyond the workspace limit is solved in chunks, which the step-by-step form below does)
    if (d.fixed_iters <= 0 && h->knobs.pipeline && !h->pipe_disabled && h->knobs.loop_async && (size_t)B <= max_rows_per_solve(h)) {
        HIP_TRY(h, hipMemsetAsync(h->d_fail, 0, 2 * sizeof(uint32_t), stream));
        A.abort_flag = h->d_fail + 1;
        hipLaunchKernelGGL(k_loop_setup, grid, block, 0, stream, A);
        h->async_loop = true;
        bool all_async = true;
        for (int i = 0; i < L && all_async; ++i) {
            rc = solve_dev_any(h, B, h->d_x0, h->d_p, nullptr, h->d_xout, h->d_status, h->d_iters, h->d_kkt, stream, nullptr, 0, nullptr);
            if (rc) { h->async_loop = false; return rc; }
            all_async = h->async_ok;             // (a batch shape the pipeline does not take: the solve has run synchronously -- start over)
            if (all_async) hipLaunchKernelGGL(k_loop_advance, dim3(B), dim3(128), 0, stream, P, A, i);
        }
        h->async_loop = false;
        if (all_async) {
            HIP_TRY(h, hipMemcpyAsync(h->h_fail, h->d_fail, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(h, hipStreamSynchronize(stream));
            replay = h->h_fail[0] != 0u || h->h_fail[1] != 0u;
            if (h->h_fail[1] != 0u) {
                h->pipe_disabled = true;
                loop_abandoned = true;
                fprintf(stderr, "[mpcgpu] closed loop: a pipeline launch was abandoned (bounded wait expired); replaying the loop with one launch per kernel\n");
            }
        }
    }
    h->loop_replayed = replay ? 1 : 0;
    if (replay) {
        A.abort_flag = nullptr;
        hipLaunchKernelGGL(k_loop_setup, grid, block, 0, stream, A);
        for (int i = 0; i < L; ++i) {
            rc = solve_dev(h, B, h->d_x0, h->d_p, nullptr, h->d_xout, h->d_status, h->d_iters, h->d_kkt, stream, nullptr, 0, nullptr);
            if (rc) return rc;
            hipLaunchKernelGGL(k_loop_advance, dim3(B), dim3(128), 0, stream, P, A, i);
        }
    }
    if (loop_abandoned && ++h->pipe_aborts < mpc_handle::PIPE_ABORTS_MAX) h->pipe_disabled = false;         // (see solve_dev_impl)
    HIP_TRY(h, hipGetLastError());
    return MPC_OK;
}

int mpc_closed_loop_batch_dev(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* d_init_state, const double* d_path,
                              const double* d_orient, const double* d_vdes, double* d_traj, double* d_ctrl, int32_t* d_step_status,
                              void* stream_) {
    return mpc_closed_loop_batch_dev_ex(h, B, L, Lp, d_init_state, d_path, d_orient, d_vdes, 0, 0.0, 0, d_traj, d_ctrl, d_step_status, stream_);
}

int mpc_closed_loop_batch_ex(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* init_state, const double* path,
                             const double* orient, const double* vdes, int32_t noise_mode, double sigma, uint64_t seed, double* traj,
                             double* ctrl, int32_t* step_status) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || L <= 0 || Lp <= 0 || !init_state || !path || !orient || !vdes || !traj || !ctrl) { h->err = "closed loop: null or empty argument"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    // staging buffers of the host-buffer form: owned by the handle, grow-only (no allocation per call once they have their size)
    const size_t nB = (size_t)B;
    double* di = static_cast<double*>(scratch_get(h, 29, nB * 5 * 8));
    double* dp = static_cast<double*>(scratch_get(h, 30, nB * Lp * 2 * 8));
    double* dor = static_cast<double*>(scratch_get(h, 31, nB * Lp * 8));
    double* dv = static_cast<double*>(scratch_get(h, 32, nB * 8));
    double* dt_ = static_cast<double*>(scratch_get(h, 33, nB * L * 5 * 8));
    double* dc = static_cast<double*>(scratch_get(h, 34, nB * L * 2 * 8));
    int32_t* ds = static_cast<int32_t*>(scratch_get(h, 35, nB * L * 4));
    if (!di || !dp || !dor || !dv || !dt_ || !dc || !ds) {
        h->err = "closed loop: out of device memory";
        return MPC_ERR_HIP;
    }
    hipStream_t s = h->own_stream;
    int rc = MPC_OK;
    if (hipMemcpyAsync(di, init_state, nB * 5 * 8, hipMemcpyHostToDevice, s) != hipSuccess || hipMemcpyAsync(dp, path, nB * Lp * 2 * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(dor, orient, nB * Lp * 8, hipMemcpyHostToDevice, s) != hipSuccess || hipMemcpyAsync(dv, vdes, nB * 8, hipMemcpyHostToDevice, s) != hipSuccess) {
        h->err = "closed loop: host to device copy failed";
        rc = MPC_ERR_HIP;
    }
    if (!rc) rc = mpc_closed_loop_batch_dev_ex(h, B, L, Lp, di, dp, dor, dv, noise_mode, sigma, seed, dt_, dc, ds, (void*)s);
    if (!rc) {
        if (hipMemcpyAsync(traj, dt_, nB * L * 5 * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipMemcpyAsync(ctrl, dc, nB * L * 2 * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
            (step_status && hipMemcpyAsync(step_status, ds, nB * L * 4, hipMemcpyDeviceToHost, s) != hipSuccess) || hipStreamSynchronize(s) != hipSuccess) {
            h->err = "closed loop: device to host copy failed";
            rc = MPC_ERR_HIP;
        }
    }
    return rc;
}

int mpc_closed_loop_batch(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* init_state, const double* path,
                          const double* orient, const double* vdes, double* traj, double* ctrl, int32_t* step_status) {
    return mpc_closed_loop_batch_ex(h, B, L, Lp, init_state, path, orient, vdes, 0, 0.0, 0, traj, ctrl, step_status);
}
int mpc_last_loop_replayed(const mpc_handle* h) { return h ? h->loop_replayed : MPC_ERR_INVALID; }

int mpc_metrics_batch_dev(mpc_handle* h, int32_t B, int32_t L, int32_t Lo, const double* d_traj, const double* d_ref_path, const double* d_origin_path,
                          double r_sum, int32_t all_pairs, double* d_deviation, double* d_rmsd, double* d_clearance, void* stream_) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || L < 2 || !d_traj || (d_deviation && (!d_origin_path || Lo <= 0))) { h->err = "metrics: B > 0, L >= 2, traj (and origin_path with deviation) are required"; return MPC_ERR_INVALID; }
    if (d_rmsd && !d_ref_path) { h->err = "metrics: rmsd needs ref_path"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream_;
    double* dob = static_cast<double*>(scratch_get(h, 3, 6 * 8));
    if (!dob) { h->err = "metrics: out of device memory"; return MPC_ERR_HIP; }
    HIP_TRY(h, hipMemcpyAsync(dob, h->hp.desc.obstacle, 6 * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_metrics, dim3(B), dim3(256), 0, s, L, Lo, d_traj, d_ref_path, d_origin_path, dob, h->hp.desc.ego_offset, r_sum, (int)all_pairs,
                       d_deviation, d_rmsd, d_clearance);
    HIP_TRY(h, hipGetLastError());
    return MPC_OK;
}

int mpc_metrics_batch(mpc_handle* h, int32_t B, int32_t L, int32_t Lo, const double* traj, const double* ref_path, const double* origin_path,
                      double r_sum, int32_t all_pairs, double* deviation, double* rmsd, double* clearance) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || L < 2 || !traj || (deviation && (!origin_path || Lo <= 0))) { h->err = "metrics: B > 0, L >= 2, traj (and origin_path with deviation) are required"; return MPC_ERR_INVALID; }
    if (rmsd && !ref_path) { h->err = "metrics: rmsd needs ref_path"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nB = (size_t)B;
    hipStream_t s = h->own_stream;
    double* dt_ = static_cast<double*>(scratch_get(h, 4, nB * L * 5 * 8));
    double* dr = rmsd ? static_cast<double*>(scratch_get(h, 5, nB * L * 2 * 8)) : nullptr;
    double* drm = rmsd ? static_cast<double*>(scratch_get(h, 6, nB * 2 * 8)) : nullptr;
    double* dorig = deviation ? static_cast<double*>(scratch_get(h, 7, nB * Lo * 2 * 8)) : nullptr;
    double* ddev = deviation ? static_cast<double*>(scratch_get(h, 8, nB * L * 8)) : nullptr;
    double* dcl = clearance ? static_cast<double*>(scratch_get(h, 9, nB * 8)) : nullptr;
    if (!dt_ || (rmsd && (!dr || !drm)) || (deviation && (!dorig || !ddev)) || (clearance && !dcl)) { h->err = "metrics: out of device memory"; return MPC_ERR_HIP; }
    HIP_TRY(h, hipMemcpyAsync(dt_, traj, nB * L * 5 * 8, hipMemcpyHostToDevice, s));
    if (rmsd) HIP_TRY(h, hipMemcpyAsync(dr, ref_path, nB * L * 2 * 8, hipMemcpyHostToDevice, s));
    if (deviation) HIP_TRY(h, hipMemcpyAsync(dorig, origin_path, nB * Lo * 2 * 8, hipMemcpyHostToDevice, s));
    const int rc = mpc_metrics_batch_dev(h, B, L, Lo, dt_, dr, dorig, r_sum, all_pairs, ddev, drm, dcl, (void*)s);
    if (rc) return rc;
    if (deviation) HIP_TRY(h, hipMemcpyAsync(deviation, ddev, nB * L * 8, hipMemcpyDeviceToHost, s));
    if (rmsd) HIP_TRY(h, hipMemcpyAsync(rmsd, drm, nB * 2 * 8, hipMemcpyDeviceToHost, s));
    if (clearance) HIP_TRY(h, hipMemcpyAsync(clearance, dcl, nB * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return MPC_OK;
}

int mpc_validity_batch_dev(mpc_handle* h, int32_t B, int32_t L, const double* d_traj, double ego_length, double ego_width, int32_t n_obst,
                           const double* d_obst, int32_t n_left, const double* d_left, int32_t n_right, const double* d_right,
                           int32_t* d_first_collision, int32_t* d_first_off_road, void* stream_) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || L <= 0 || !d_traj || !d_first_collision || !d_first_off_road || n_obst < 0 || (n_obst > 0 && !d_obst) ||
        (n_left > 0 && !d_left) || (n_right > 0 && !d_right) || !(ego_length > 0.0) || !(ego_width > 0.0)) {
        h->err = "validity: B, L > 0, traj, both outputs and consistent obstacle / boundary arguments are required";
        return MPC_ERR_INVALID;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(k_validity, dim3(B), dim3(128), 0, (hipStream_t)stream_, L, d_traj, ego_length, ego_width, n_obst, d_obst, n_left, d_left, n_right,
                       d_right, d_first_collision, d_first_off_road);
    HIP_TRY(h, hipGetLastError());
    return MPC_OK;
}

int mpc_validity_batch(mpc_handle* h, int32_t B, int32_t L, const double* traj, double ego_length, double ego_width, int32_t n_obst,
                       const double* obst, int32_t n_left, const double* left, int32_t n_right, const double* right,
                       int32_t* first_collision, int32_t* first_off_road) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || L <= 0 || !traj || !first_collision || !first_off_road) { h->err = "validity: B, L > 0, traj and both outputs are required"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = h->own_stream;
    const size_t nB = (size_t)B;
    double* dt_ = static_cast<double*>(scratch_get(h, 4, nB * L * 5 * 8));
    double* dob = n_obst > 0 ? static_cast<double*>(scratch_get(h, 5, (size_t)n_obst * L * 5 * 8)) : nullptr;
    double* dl = n_left > 0 ? static_cast<double*>(scratch_get(h, 6, (size_t)n_left * 2 * 8)) : nullptr;
    double* dr = n_right > 0 ? static_cast<double*>(scratch_get(h, 7, (size_t)n_right * 2 * 8)) : nullptr;
    int32_t* dc = static_cast<int32_t*>(scratch_get(h, 11, nB * 4));
    int32_t* dor = static_cast<int32_t*>(scratch_get(h, 12, nB * 4));
    if (!dt_ || (n_obst > 0 && !dob) || (n_left > 0 && !dl) || (n_right > 0 && !dr) || !dc || !dor) { h->err = "validity: out of device memory"; return MPC_ERR_HIP; }
    HIP_TRY(h, hipMemcpyAsync(dt_, traj, nB * L * 5 * 8, hipMemcpyHostToDevice, s));
    if (n_obst > 0) HIP_TRY(h, hipMemcpyAsync(dob, obst, (size_t)n_obst * L * 5 * 8, hipMemcpyHostToDevice, s));
    if (n_left > 0) HIP_TRY(h, hipMemcpyAsync(dl, left, (size_t)n_left * 2 * 8, hipMemcpyHostToDevice, s));
    if (n_right > 0) HIP_TRY(h, hipMemcpyAsync(dr, right, (size_t)n_right * 2 * 8, hipMemcpyHostToDevice, s));
    const int rc = mpc_validity_batch_dev(h, B, L, dt_, ego_length, ego_width, n_obst, dob, n_left, dl, n_right, dr, dc, dor, (void*)s);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(first_collision, dc, nB * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(first_off_road, dor, nB * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return MPC_OK;
}

int mpc_forces_stage_eval(mpc_handle* h, int32_t B, int32_t terminal, const double* z, const double* p, double* f, double* grad_f,
                          double* c, double* jac_c, double* hval, double* jac_h) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || !z || !p) { h->err = "forces stage eval: B > 0, z and p are required"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    const mpc_problem_desc& d = h->hp.desc;
    const size_t nB = (size_t)B;
    const size_t sz[8] = {nB * 7, nB * 10, nB, nB * 7, nB * 5, nB * 35, nB * 10, nB * 70};
    double* host[8] = {const_cast<double*>(z), const_cast<double*>(p), f, grad_f, terminal ? nullptr : c, terminal ? nullptr : jac_c, hval, jac_h};
    double* dev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t s = h->own_stream;
    bool ok = true;
    for (int i = 0; i < 8 && ok; ++i)
        if (host[i]) ok = hipMalloc(&dev[i], sz[i] * sizeof(double)) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) ok = hipMemcpyAsync(dev[i], host[i], sz[i] * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok) {
        ForcesArgs A{};
        A.B = B; A.terminal = terminal ? 1 : 0;
        A.dt = d.dt; A.l = d.wheelbase; A.wb = d.friction_div; A.rho = d.ego_offset;
        for (int i = 0; i < 5; ++i) { A.Q[i] = d.Q[i]; A.Pt[i] = d.P[i]; }
        A.R[0] = d.R[0]; A.R[1] = d.R[1];
        A.z = dev[0]; A.p = dev[1]; A.f = dev[2]; A.grad_f = dev[3]; A.c = dev[4]; A.jac_c = dev[5]; A.h = dev[6]; A.jac_h = dev[7];
        hipLaunchKernelGGL(k_forces_stage, dim3((B + 127) / 128), dim3(128), 0, s, A);
        for (int i = 2; i < 8 && ok; ++i)
            if (host[i]) ok = hipMemcpyAsync(host[i], dev[i], sz[i] * sizeof(double), hipMemcpyDeviceToHost, s) == hipSuccess;
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
    }
    for (int i = 0; i < 8; ++i) (void)hipFree(dev[i]);
    if (!ok) { h->err = "forces stage eval: HIP allocation, copy or launch failed"; return MPC_ERR_HIP; }
    return MPC_OK;
}

int mpc_forces_solve_batch_dev(mpc_handle* h, int32_t B, const double* d_x0, const double* d_xinit, const double* d_all_parameters,
                               const double* lb, const double* ub, const double* hl, const double* hu, int32_t hessian_mode,
                               double* d_x_out, int32_t* d_exitflag, int32_t* d_it, double* d_res, void* stream_) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || !d_x0 || !d_xinit || !d_all_parameters || !lb || !ub || !hl || !hu || !d_x_out) { h->err = "forces solve: null or empty argument"; return MPC_ERR_INVALID; }
    const mpc_problem_desc& d = h->hp.desc;
    if (d.nx != 5) { h->err = "the FORCES formulation has 5 states (z = [deltaDot, aLong, x, y, delta, v, psi])"; return MPC_ERR_INVALID; }
    // (k_forces_qp holds all stages of an instance in one workgroup of at most 256 threads, one thread per stage)
    if (d.N > 192) { h->err = "forces solve: horizons above 192 stages are not supported (one thread per stage, 192 threads per workgroup: their stage rows live in LDS)"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    const int N = d.N;
    const size_t nB = (size_t)B, Bp = (nB + 63) / 64 * 64;
    hipStream_t s = (hipStream_t)stream_;
    if ((size_t)FQ_ROWS * N * Bp * 8 >= ((size_t)1 << 32)) { h->err = "forces solve: batch too large for one call (workspace of 4 GiB)"; return MPC_ERR_INVALID; }
    double* dws = static_cast<double*>(scratch_get(h, 10, (size_t)FQ_ROWS * N * Bp * 8));
    int32_t* dflag = d_exitflag ? d_exitflag : static_cast<int32_t*>(scratch_get(h, 11, nB * 4));
    int32_t* dit = d_it ? d_it : static_cast<int32_t*>(scratch_get(h, 12, nB * 4));
    double* dres = d_res ? d_res : static_cast<double*>(scratch_get(h, 13, nB * 8));
    if (!dws || !dflag || !dit || !dres) { h->err = "forces solve: out of device memory"; return MPC_ERR_HIP; }
    ForcesQpArgs A{};
    A.B = B; A.Bp = (int32_t)Bp; A.N = N; A.max_it = 60;
    A.dt = d.dt; A.l = d.wheelbase; A.wb = d.friction_div; A.rho = d.ego_offset;
    A.tol = 1e-4; A.tol_mu = 1e-6;
    for (int i = 0; i < 5; ++i) { A.Q[i] = d.Q[i]; A.Pt[i] = d.P[i]; }
    A.R[0] = d.R[0]; A.R[1] = d.R[1];
    forces_hessian_diag(hessian_mode, A.Q, A.R, A.Pt, A.hd, A.hdN);
    for (int i = 0; i < 7; ++i) { A.lb[i] = lb[i]; A.ub[i] = ub[i]; }
    for (int i = 0; i < 10; ++i) { A.hl[i] = hl[i]; A.hu[i] = hu[i]; }
    A.zbar = d_x0; A.params = d_all_parameters; A.xinit = d_xinit; A.z_out = d_x_out; A.iters = dit; A.status = dflag; A.kkt = dres; A.ws = dws; A.ws_bytes = (uint32_t)((size_t)FQ_ROWS * N * Bp * 8);
    // instances per workgroup: all N stages of IB instances in one workgroup whose LDS rows ([4 + FL_ROWS][threads] doubles: the RK4 Jacobian,
    // cost-to-go, hand-over rows and the Jacobian of h) fit a CU: at most 192 threads
    constexpr int FQ_MAX_THREADS = 192;
    int IB = 1;
    while (((IB * 2 * N + 63) / 64) * 64 <= FQ_MAX_THREADS && IB < 64) IB *= 2;
    const int threads = ((IB * N + 63) / 64) * 64;
    if ((size_t)(4 + FL_ROWS) * threads * sizeof(double) > 160 * 1024 - 1024) { h->err = "forces solve: the horizon is too long for the stage rows of one instance in LDS"; return MPC_ERR_INVALID; }
    if (!h->attr_set_fq) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forces_qp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - 1024)));
        h->attr_set_fq = true;
    }
    hipLaunchKernelGGL(k_forces_qp, dim3((B + IB - 1) / IB), dim3(threads), (size_t)(4 + FL_ROWS) * threads * sizeof(double), s, A, IB);
    HIP_TRY(h, hipGetLastError());
    return MPC_OK;
}

int mpc_forces_solve_batch(mpc_handle* h, int32_t B, const double* x0, const double* xinit, const double* all_parameters,
                           const double* lb, const double* ub, const double* hl, const double* hu, int32_t hessian_mode,
                           double* x_out, int32_t* exitflag, int32_t* it, double* res) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || !x0 || !xinit || !all_parameters || !lb || !ub || !hl || !hu || !x_out) { h->err = "forces solve: null or empty argument"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t N = (size_t)h->hp.desc.N, nB = (size_t)B;
    hipStream_t s = h->own_stream;
    double* dz = static_cast<double*>(scratch_get(h, 4, nB * N * 7 * 8));
    double* dxi = static_cast<double*>(scratch_get(h, 5, nB * 5 * 8));
    double* dpar = static_cast<double*>(scratch_get(h, 6, nB * N * 10 * 8));
    double* dout = static_cast<double*>(scratch_get(h, 7, nB * N * 7 * 8));
    double* dres = static_cast<double*>(scratch_get(h, 13, nB * 8));
    int32_t* dflag = static_cast<int32_t*>(scratch_get(h, 11, nB * 4));
    int32_t* dit = static_cast<int32_t*>(scratch_get(h, 12, nB * 4));
    if (!dz || !dxi || !dpar || !dout || !dres || !dflag || !dit) { h->err = "forces solve: out of device memory"; return MPC_ERR_HIP; }
    HIP_TRY(h, hipMemcpyAsync(dz, x0, nB * N * 7 * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(dxi, xinit, nB * 5 * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(dpar, all_parameters, nB * N * 10 * 8, hipMemcpyHostToDevice, s));
    const int rc = mpc_forces_solve_batch_dev(h, B, dz, dxi, dpar, lb, ub, hl, hu, hessian_mode, dout, dflag, dit, dres, (void*)s);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(x_out, dout, nB * N * 7 * 8, hipMemcpyDeviceToHost, s));
    if (exitflag) HIP_TRY(h, hipMemcpyAsync(exitflag, dflag, nB * 4, hipMemcpyDeviceToHost, s));
    if (it) HIP_TRY(h, hipMemcpyAsync(it, dit, nB * 4, hipMemcpyDeviceToHost, s));
    if (res) HIP_TRY(h, hipMemcpyAsync(res, dres, nB * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return MPC_OK;
}

int mpc_forces_closed_loop_batch_dev(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* d_init_state, const double* d_init_acc,
                                     const double* d_path, const double* d_orient, const double* d_vdes, const double* lb, const double* ub,
                                     const double* hl, const double* hu, int32_t hessian_mode, int32_t noise_mode, double sigma, uint64_t seed,
                                     double* d_traj, double* d_ctrl, int32_t* d_step_flag, void* stream_) {
    if (!h) return MPC_ERR_INVALID;
    const mpc_problem_desc& d = h->hp.desc;
    if (B <= 0 || L <= 0 || Lp <= 0 || L < d.N || !d_init_state || !d_path || !d_orient || !d_vdes || !d_traj || !d_ctrl || !lb || !ub || !hl || !hu) {
        h->err = "forces closed loop: B, L, Lp > 0, L >= N and init_state, path, orient, vdes, traj, ctrl, lb, ub, hl, hu are required";
        return MPC_ERR_INVALID;
    }
    if (!(noise_mode == 0 || noise_mode == 2) || (noise_mode && !(sigma >= 0.0))) { h->err = "forces closed loop: noise_mode 0 or 2, sigma >= 0"; return MPC_ERR_INVALID; }
    if (d.nx != 5) { h->err = "the FORCES formulation has 5 states"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nB = (size_t)B, N = (size_t)d.N;
    double* st = static_cast<double*>(scratch_get(h, 16, nB * 5 * 8));
    double* zb = static_cast<double*>(scratch_get(h, 17, nB * N * 7 * 8));
    double* par = static_cast<double*>(scratch_get(h, 18, nB * N * 10 * 8));
    double* zo = static_cast<double*>(scratch_get(h, 19, nB * N * 7 * 8));
    int32_t* fl = static_cast<int32_t*>(scratch_get(h, 20, nB * 4));
    if (!st || !zb || !par || !zo || !fl) { h->err = "forces closed loop: out of device memory"; return MPC_ERR_HIP; }
    hipStream_t s = (hipStream_t)stream_;
    ForcesLoopArgs A{};
    A.B = B; A.N = d.N; A.L = L; A.Lp = Lp;
    A.init_state = d_init_state; A.init_acc = d_init_acc; A.path = d_path; A.orient = d_orient; A.vdes = d_vdes;
    for (int i = 0; i < 6; ++i) A.obstacle[i] = d.obstacle[i];
    A.state = st; A.zbar = zb; A.params = par; A.z_out = zo; A.exitflag = fl;
    A.traj = d_traj; A.ctrl = d_ctrl; A.step_flag = d_step_flag;
    A.dt = d.dt; A.wheelbase = d.wheelbase;
    A.noise_mode = noise_mode; A.sigma = sigma; A.seed_lo = (uint32_t)seed; A.seed_hi = (uint32_t)(seed >> 32);
    const dim3 grid((B + 127) / 128), block(128);
    hipLaunchKernelGGL(k_floop_setup, grid, block, 0, s, A);
    for (int k = 0; k < L; ++k) {                       // nothing comes back to the host between the steps
        hipLaunchKernelGGL(k_floop_params, grid, block, 0, s, A, k);
        const int rc = mpc_forces_solve_batch_dev(h, B, zb, st, par, lb, ub, hl, hu, hessian_mode, zo, fl, nullptr, nullptr, (void*)s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_floop_advance, grid, block, 0, s, A, k);
    }
    HIP_TRY(h, hipGetLastError());
    return MPC_OK;
}

int mpc_forces_closed_loop_batch(mpc_handle* h, int32_t B, int32_t L, int32_t Lp, const double* init_state, const double* init_acc, const double* path,
                                 const double* orient, const double* vdes, const double* lb, const double* ub, const double* hl, const double* hu,
                                 int32_t hessian_mode, int32_t noise_mode, double sigma, uint64_t seed, double* traj, double* ctrl, int32_t* step_flag) {
    if (!h) return MPC_ERR_INVALID;
    if (B <= 0 || L <= 0 || Lp <= 0 || !init_state || !path || !orient || !vdes || !traj || !ctrl) { h->err = "forces closed loop: null or empty argument"; return MPC_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nB = (size_t)B;
    hipStream_t s = h->own_stream;
    double* di = static_cast<double*>(scratch_get(h, 21, nB * 5 * 8));
    double* da = static_cast<double*>(scratch_get(h, 22, nB * 8));
    double* dp = static_cast<double*>(scratch_get(h, 23, nB * Lp * 2 * 8));
    double* dor = static_cast<double*>(scratch_get(h, 24, nB * Lp * 8));
    double* dv = static_cast<double*>(scratch_get(h, 25, nB * 8));
    double* dt_ = static_cast<double*>(scratch_get(h, 26, nB * L * 5 * 8));
    double* dc = static_cast<double*>(scratch_get(h, 27, nB * L * 2 * 8));
    int32_t* df = static_cast<int32_t*>(scratch_get(h, 28, nB * L * 4));
    if (!di || !da || !dp || !dor || !dv || !dt_ || !dc || !df) { h->err = "forces closed loop: out of device memory"; return MPC_ERR_HIP; }
    HIP_TRY(h, hipMemcpyAsync(di, init_state, nB * 5 * 8, hipMemcpyHostToDevice, s));
    if (init_acc) HIP_TRY(h, hipMemcpyAsync(da, init_acc, nB * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(dp, path, nB * Lp * 2 * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(dor, orient, nB * Lp * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(dv, vdes, nB * 8, hipMemcpyHostToDevice, s));
    const int rc = mpc_forces_closed_loop_batch_dev(h, B, L, Lp, di, init_acc ? da : nullptr, dp, dor, dv, lb, ub, hl, hu, hessian_mode, noise_mode, sigma, seed,
                                                    dt_, dc, step_flag ? df : nullptr, (void*)s);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(traj, dt_, nB * L * 5 * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(ctrl, dc, nB * L * 2 * 8, hipMemcpyDeviceToHost, s));
    if (step_flag) HIP_TRY(h, hipMemcpyAsync(step_flag, df, nB * L * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return MPC_OK;
}

}  // extern "C"
