// mpc_riccati_mfma.h -- the Riccati factor / solve of ONE instance by ONE wavefront on the fp64 matrix pipe (v_mfma_f64_4x4x4_f64), the
// KKT solve of the workgroup-resident solve path (k_solve_wg in mpcgpu.hip).  Product code, device only.
//
// Why a second formulation.  riccati_tile runs one instance per LANE: 64 instances per wavefront, but a 31-stage chain of ~2 200 ticks
// per stage whatever the number of live lanes -- 31 us of latency per interior-point iteration.  Here the augmented cost-to-go
//     M_k = [ P_k  p_k ; p_k'  * ]   (8 x 8: states 0..NX-1, zero rows up to 6, the affine coordinate at index 7)
// lies ACROSS the 64 lanes of a wavefront and one backward stage is five matrix instructions,
//     Y = M+ At          T = At' Y + Ht          M = T + Gt' Kt (+ delta_w I)          with  Gt = B'Y + [Hux | gu],  Kt = -Lam^-1 Gt
// (At = [A -c; 0 1], Ht = [H gx; gx' 0]: the affine terms ride along as row / column 7, so p_k and k_ff need no recursion of their
// own): 602 ticks per stage instead of 2 230, issue bound (~95 instructions, a wavefront issues in order).  The forward sweep is ONE
// matrix instruction per stage: x~' = (At + Bt Kt) x~ with the two rows of Kt in the spare rows 6, 7 of the same product (150 ticks
// against 720).
//
// Lane map of v_mfma_f64_4x4x4_f64 (4 blocks of 4x4x4, one double per lane and operand; measured with tools/ubench/mfma_probe.hip, the
// guide documents the 16x16x4 form only): lane L = 16 x + 4 blk + y,
//     A operand: A_blk[i = y][k = x]      B operand: B_blk[k = x][j = y]      C / D: D_blk[i = x][j = y].
// An 8 x 8 matrix is 2 x 2 blocks, blk = 2 hi + lo: "natural" (= D) layout M[4 hi + x][4 lo + y], i.e. row R = 4 * bit3(L) + (L >> 4),
// column C = L & 7.  A D register used as A operand is read TRANSPOSED block by block; M+ is symmetric, so as A operand its block
// (hi, lo) reads as M_{lo, hi}.  Operands that come out of a product are re-arranged by moving 4-lane banks inside each 16-lane row
// (DPP row shifts / rotations with a bank mask, no LDS); operands that come from the stage data (At, Ht) are read from the LDS record
// of the stage with per-lane offsets -- every lane knows which entry of the sparse At it feeds.  Per stage:
//     Y      one instruction per column block J: block (hi, lo) = M_{lo, hi} At_{hi, J}; the sum over hi (DPP row rotation by 8 + add)
//            leaves Y_{lo, J} in both halves of every row -- two independent instructions, no chain
//     T      two chained instructions (k-steps); their B operands are the banks of the two Y registers merged by one DPP move each
//     Gt     rows 2, 3 of Y (the first of those B operands) brought down to rows 0, 1 by v_permlane32_swap, replicated over hi by
//            construction: Kt = -Lam^-1 Gt (partner row by ds_swizzle, Lam^-1 by every lane from three v_readlane of M+) IS the B
//            operand of the rank-2 update, Gt' needs one bank move
//     M      one instruction: T + Gt' Kt
// (tools/ubench/ric_mfma_test.hip checks both sweeps against the scalar recursion; DESIGN.md section 4 has the history of the formulation.)
//
// Reference: the linear solve inside IPOPT's step computation for the NLP of MPC_Planner/optimizer.py:373-558 (the reference hands it
// to MUMPS); recursion and inertia-correction schedule as riccati_instance (mpc_stage_math.h).  The rounding differs from the
// one-instance-per-lane recursion (dense 8 x 8 products instead of the sparse update): results agree to ~1e-13 relative.
#pragma once
#include <type_traits>
#include "mpc_stage_math.h"

namespace mpc {

// (the LDS record of one (instance, stage) -- stage block, gains, cost-to-go, step -- is Rec<NX> of mpc_stage_math.h)

#if defined(__HIP_DEVICE_COMPILE__)

// ---- cross-lane primitives on doubles ---------------------------------------------------------------------------------------------
template <int CTRL, int BANKS>
__device__ __forceinline__ double wv_dpp(double old, double src) {
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANKS, false);
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANKS, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_SHL4 = 0x104, DPP_SHL8 = 0x108, DPP_SHR4 = 0x114, DPP_SHR8 = 0x118;     // row_shl: lane <- lane + n, row_shr: lane <- lane - n
// lanes 0..31 <- lanes 32..63 of v, lanes 32..63 <- 0
__device__ __forceinline__ double wv_upper_to_lower(double v) {
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), 0u, false, false);
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), 0u, false, false);
    return __hiloint2double((int)h[1], (int)l[1]);
}
// 16-lane rows swapped inside each half of the wavefront
__device__ __forceinline__ double wv_swap16(double v) {
    return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F), __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F));
}
__device__ __forceinline__ double wv_readlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wv_bpermute(double v, int byte_addr) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v)));
}
__device__ __forceinline__ double wv_mfma(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// ---- per-lane constants: which entries of the stage record this lane feeds into which operand ------------------------------------------
// lane L = 16 x + 8 hi + 4 lo + y  (hardware block (hi, lo))
template <int NX>
struct MfmaLane {
    int oB[2], oAA[2], oHC, oHA;        // backward: At as B operand of Y (per block column J) / as transposed A operand of T (per k-step), Ht, [Hux | gu]
    int oF1, oF2;                       // forward: A operand = rec[oF1] + fscale * rec[oF2]
    double fscale, dmask;               // dmask: 1 on the diagonal of the state block (delta_w goes there)
    int pk_row;                         // entry of the cost-to-go this lane stores (P_k upper triangle, p_k: slot Rec::PK + pk_row of the stage's record), or -1
    int tr_addr;                        // byte address of the transposed lane (ds_bpermute), for the symmetrisation
    int k_off;                          // backward: where this lane writes its entry of Kt (or the dummy slot)
    int dz_row;                         // forward: entry of the step this lane stores (du: 0, 1 -> Rec::DU; dx: 2 + R -> Rec::DX of the NEXT record), or -1
    bool dz_next;                       // ... of stage k + 1 (the dx lanes)
    int fix;                            // forward: 0 = keep the product, 1 = coordinate forced to 0 (spare row), 2 = forced to 1 (affine coordinate)
    int R, C, Rb;                       // natural row / column; row of x~ this lane carries as B operand of the forward sweep
};
#endif   // __HIP_DEVICE_COMPILE__

// The constants depend on the lane (and NX) alone: a table built by the compiler, four packed words per lane.  (They used to be worked out
// by every wavefront in every round -- ~900 instructions of comparisons and exec-mask branches, 3 - 4 k ticks of a 59 k-tick round -- so as
// not to hold twenty registers across the stage phases; the four words stay in registers and unpack in ~30 instructions.)
template <int NX>
MPC_HD constexpr int rec_aoff(int r, int c) {
    using RC = Rec<NX>;
    if (r < NX && c < NX) {
        if (r == c) return RC::ONE;
        if (r == 0 && c == 3) return RC::A + 0;
        if (r == 0 && c == 4) return RC::A + 1;
        if (r == 1 && c == 3) return RC::A + 2;
        if (r == 1 && c == 4) return RC::A + 3;
        if (r == 4 && c == 2) return RC::A + 4;
        if (r == 4 && c == 3) return RC::A + 5;
        if (NX == 6 && r == 5 && c == 3) return RC::DT;
        return RC::ZERO;
    }
    if (r < NX && c == 7) return RC::NCN + r;
    if (r == 7 && c == 7) return RC::ONE;
    return RC::ZERO;
}
template <int NX>
MPC_HD constexpr int rec_hoff(int r, int c) {
    using RC = Rec<NX>;
    if (r < NX && c < NX) {
        const int i = r < c ? r : c, j = r < c ? c : r;
        const int hr = i == j ? i : (i == 0 && j == 1) ? NX : (i == 0 && j == 4) ? NX + 1 : (i == 1 && j == 4) ? NX + 2 : (i == 2 && j == 3) ? NX + 3
                     : (i == 3 && j == 4) ? NX + 4 : -1;                                          // (Dim::hrow with run-time arguments)
        return hr >= 0 ? RC::H + hr : RC::ZERO;
    }
    if (r < NX && c == 7) return RC::GX + r;
    if (r == 7 && c < NX) return RC::GX + c;
    return RC::ZERO;
}
struct MfmaWords { uint32_t w[4]; };
// w0: oB0 | oB1 << 8 | oAA0 << 16 | oAA1 << 24      w1: oHC | oHA << 8 | oF1 << 16 | oF2 << 24
// w2: (pk_row + 1) | (k_off + 1) << 6 | (dz_row + 1) << 13 | dz_next << 17 | fix << 18 | R << 20 | C << 23 | Rb << 26 | (fscale is dt) << 29 | dmask << 30
// w3: tr_addr
template <int NX>
MPC_HD constexpr MfmaWords mfma_lane_words(int lane) {
    using RC = Rec<NX>;
    using D = Dim<NX>;
    static_assert(RC::SIZE < 256, "record offsets are packed into bytes");
    const int x = lane >> 4, hi = (lane >> 3) & 1, lo = (lane >> 2) & 1, y = lane & 3;
    const int R = 4 * hi + x, C = lane & 7, Rb = 4 * hi + x;
    const int oB0 = rec_aoff<NX>(4 * hi + x, y), oB1 = rec_aoff<NX>(4 * hi + x, 4 + y);        // block (hi, lo) of the B operand of column block q: At_{hi, q}
    const int oAA0 = rec_aoff<NX>(x, 4 * hi + y), oAA1 = rec_aoff<NX>(4 + x, 4 * hi + y);      // k-step q: (At')_{hi, q} read transposed
    const int oHC = rec_hoff<NX>(R, C);
    // rows x = 0, 1 of Gt live in both halves (hi) of the 16-lane rows: [Hux | gu] is added there
    const int oHA = (x < 2 && C == 7) ? RC::GU + x : (x == 1 && C == 2) ? RC::HX : (x == 1 && C == 3) ? RC::HX + 1 : RC::ZERO;
    const int dmask = (R == C && R < NX) ? 1 : 0;
    int pk_row = -1;
    if (R < NX && C < NX && R <= C) pk_row = D::sidx(R, C);
    if (R < NX && C == 7) pk_row = D::NS + R;
    const int tr_addr = (16 * y + 8 * lo + 4 * hi + x) * 4;
    const int k_off = (x < 2 && hi == 0) ? ((x == 0 ? RC::K0 : RC::K1) + C) : RC::DUMMY;
    // forward sweep: hardware block (hi = K, lo = I) holds Acl'_{I K}, read transposed: A_blk[i = y][k = x] = Acl'[4 lo + y][4 hi + x]
    int oF1 = RC::ZERO, oF2 = RC::ZERO, fdt = 0;
    {
        const int r = 4 * lo + y, c = 4 * hi + x;
        if (r < NX) { oF1 = rec_aoff<NX>(r, c); oF2 = (r == 2) ? RC::K0 + c : (r == 3) ? RC::K1 + c : RC::ZERO; fdt = 1; }
        else { oF1 = RC::ZERO; oF2 = (r == 6) ? RC::K0 + c : (r == 7) ? RC::K1 + c : RC::ZERO; fdt = 0; }
    }
    // the product of the forward sweep, summed over hi: lane (x, *, lo, y = 0) carries row 4 lo + x: states, du_0 (row 6), du_1 (row 7)
    int dz_row = -1, dz_next = 0, fix = 0;
    {
        const int r = 4 * lo + x;
        if (y == 0 && hi == 0) {
            if (r < NX) { dz_row = 2 + r; dz_next = 1; }
            else if (r >= 6) dz_row = r - 6;
        }
        if (r == 7) fix = 2; else if (r >= NX) fix = 1;
        if (y != 0) fix = 1;
    }
    MfmaWords o{};
    o.w[0] = (uint32_t)oB0 | (uint32_t)oB1 << 8 | (uint32_t)oAA0 << 16 | (uint32_t)oAA1 << 24;
    o.w[1] = (uint32_t)oHC | (uint32_t)oHA << 8 | (uint32_t)oF1 << 16 | (uint32_t)oF2 << 24;
    o.w[2] = (uint32_t)(pk_row + 1) | (uint32_t)(k_off + 1) << 6 | (uint32_t)(dz_row + 1) << 13 | (uint32_t)dz_next << 17 | (uint32_t)fix << 18 | (uint32_t)R << 20 |
             (uint32_t)C << 23 | (uint32_t)Rb << 26 | (uint32_t)fdt << 29 | (uint32_t)dmask << 30;
    o.w[3] = (uint32_t)tr_addr;
    return o;
}
template <int NX> struct MfmaTab { MfmaWords lane[64]; };
template <int NX>
MPC_HD constexpr MfmaTab<NX> mfma_make_tab() {
    MfmaTab<NX> t{};
    for (int l = 0; l < 64; ++l) t.lane[l] = mfma_lane_words<NX>(l);
    return t;
}
#if defined(__HIPCC__)
template <int NX> __device__ const MfmaTab<NX> g_mfma_tab = mfma_make_tab<NX>();
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// the four words of a lane (kept in registers by the caller from round to round)
template <int NX>
__device__ __forceinline__ MfmaWords mfma_lane_load(int lane) {
    const uint4 v = *reinterpret_cast<const uint4*>(&g_mfma_tab<NX>.lane[lane & 63]);
    return MfmaWords{{v.x, v.y, v.z, v.w}};
}
template <int NX>
__device__ __forceinline__ void mfma_lane_setup(MfmaLane<NX>& m, const MfmaWords& w, double dt) {
    m.oB[0] = (int)(w.w[0] & 255u); m.oB[1] = (int)((w.w[0] >> 8) & 255u); m.oAA[0] = (int)((w.w[0] >> 16) & 255u); m.oAA[1] = (int)(w.w[0] >> 24);
    m.oHC = (int)(w.w[1] & 255u); m.oHA = (int)((w.w[1] >> 8) & 255u); m.oF1 = (int)((w.w[1] >> 16) & 255u); m.oF2 = (int)(w.w[1] >> 24);
    const uint32_t f = w.w[2];
    m.pk_row = (int)(f & 63u) - 1;
    m.k_off = (int)((f >> 6) & 127u) - 1;
    m.dz_row = (int)((f >> 13) & 15u) - 1;
    m.dz_next = ((f >> 17) & 1u) != 0u;
    m.fix = (int)((f >> 18) & 3u);
    m.R = (int)((f >> 20) & 7u); m.C = (int)((f >> 23) & 7u); m.Rb = (int)((f >> 26) & 7u);
    m.fscale = ((f >> 29) & 1u) ? dt : 1.0;
    m.dmask = ((f >> 30) & 1u) ? 1.0 : 0.0;
    m.tr_addr = (int)w.w[3];
}

// What the sweeps need to know about the instance (wave-uniform)
struct MfmaInst {
    uint32_t inst;             // instance
    double delta_last;
    bool sym_hint;             // keep the cost-to-go symmetric whatever delta_last says (IS_ILL: a circle row with a large weight)
};

// x + (x with the two halves of every 16-lane row exchanged): the sum over the block index hi, left in both halves
__device__ __forceinline__ double wv_sum_hi(double v) {
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x128, 0xF, 0xF, false);       // row_ror:8
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x128, 0xF, 0xF, false);
    return v + __hiloint2double(hi, lo);
}

// ---- backward sweeps of NI instances of one wavefront, interleaved (independent dependency chains in one instruction stream: the
// wavefront issues in order, so a second instance fills the latencies of the first).  ok[q] = false: no admissible delta_w (status -7).
// rec[q]: LDS records of instance q, stage k at rec[q] + k * Rec::SIZE; one more record in FRONT of every instance's records must be
// readable (the operand prefetch of stage 0 reads "stage -1").  dump: 64 doubles of LDS the lanes without an entry may write.  Writes
// the Kt rows and the cost-to-go P_k / p_k into the records -- the cost-to-go over the Ruu / gu / gx / H entries of the stage, which the
// sweep has in registers by then (Rec) -- nothing goes to memory.  rebuild(): called before a sweep is REPEATED with a larger delta_w;
// must put the stage blocks back into the records of all NI instances.
//
// One stage, with M = M+ in natural layout (symmetric: as A operand its block (hi, lo) is read as M_{lo, hi}):
//   column block J of Y = M+ At:  D = mfma(M, At_{hi, J}) gives M_{lo, hi} At_{hi, J} in block (hi, lo); the sum over hi (one DPP row
//                                 rotation + add) leaves Y_{lo, J} in both halves -- S_J, two independent instructions instead of a chain
//   B operands of T = At' Y + Ht: k-step K needs Y_{K, lo}: banks of S_0 / S_1 merged by one DPP each
//   Gt rows come down from rows 2, 3 of the first of them (v_permlane32_swap) already replicated over hi, so Kt = -Lam^-1 Gt is the B
//   operand of the rank-2 update as it stands and Gt' needs ONE bank move
template <int NX, int NI, class Rebuild>
__device__ __forceinline__ void mfma_backward(const PRef& P, const MfmaLane<NX>& m, const MfmaInst (&in)[NI], const mpc_lds_ptr (&rec)[NI],
                                              int lane, mpc_lds_ptr dump, double (&delta)[NI], bool (&ok)[NI], uint32_t& sweeps, Rebuild&& rebuild) {
    using RC = Rec<NX>;
    using D = Dim<NX>;
    const int N = P.N;
    const double dt = P.dt, dt2 = dt * dt;
    const int x = lane >> 4;
    // where a lane writes its entry of Kt / of the cost-to-go, as an offset from the records of the instance: the writers walk down the
    // stages, the others stay on their own double of the dump area (a shared dummy address would serialise them on one LDS bank)
    const bool writer = m.k_off != RC::DUMMY, pwriter = m.pk_row >= 0;
    const int w_inc = writer ? RC::SIZE : 0, p_inc = pwriter ? RC::SIZE : 0;
    int w_first[NI], p_first[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        w_first[q] = writer ? N * RC::SIZE + m.k_off : (int)(dump - rec[q]) + lane;
        p_first[q] = pwriter ? N * RC::SIZE + RC::PK + m.pk_row : (int)(dump - rec[q]) + lane;
    }
    bool need[NI], sym[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        delta[q] = 0.0;
        need[q] = true;
        ok[q] = false;
        sym[q] = in[q].delta_last != 0.0 || in[q].sym_hint;
    }
    for (bool first = true;; first = false) {
        if (!first) rebuild();
        ++sweeps;
        bool symm = false, symq[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) { symq[q] = sym[q] || delta[q] != 0.0; symm = symm || symq[q]; }
        double M[NI], b0[NI], b1[NI], aa0[NI], aa1[NI], hc[NI], ha[NI], ruu0[NI], ruu1[NI];
        mpc_lds_ptr pb0[NI], pb1[NI], paa0[NI], paa1[NI], phc[NI], pha[NI], pruu[NI];
        bool good[NI];
        int wdec = 0, pdec = 0;
        // terminal stage: M_N = Ht_N + delta_w I
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            mpc_lds_ptr r = rec[q] + N * RC::SIZE;
            M[q] = r[m.oHC] + delta[q] * m.dmask;
            good[q] = true;
            rec[q][p_first[q]] = M[q];                   // (over the terminal stage's own H / gx: every lane has read its entry)
            r = rec[q] + (N - 1) * RC::SIZE;
            b0[q] = r[m.oB[0]]; b1[q] = r[m.oB[1]]; aa0[q] = r[m.oAA[0]]; aa1[q] = r[m.oAA[1]]; hc[q] = r[m.oHC]; ha[q] = r[m.oHA];
            ruu0[q] = r[RC::RUU]; ruu1[q] = r[RC::RUU + 1];
            // operand addresses of the lane, kept on the LOWER record of the two a loop trip prefetches (the other one is an immediate
            // offset away): seven address updates per two stages instead of one per read
            r = rec[q] + (N - 3) * RC::SIZE;
            pb0[q] = r + m.oB[0]; pb1[q] = r + m.oB[1]; paa0[q] = r + m.oAA[0]; paa1[q] = r + m.oAA[1]; phc[q] = r + m.oHC; pha[q] = r + m.oHA;
            pruu[q] = r + RC::RUU;
            // (as whole addresses in vector registers: the compiler would otherwise keep the uniform part aside and add it at every read)
            asm volatile("" : "+v"(pb0[q]), "+v"(pb1[q]), "+v"(paa0[q]), "+v"(paa1[q]), "+v"(phc[q]), "+v"(pha[q]), "+v"(pruu[q]));
        }
        auto bump = [&](int by) {
#pragma unroll
            for (int q = 0; q < NI; ++q) { pb0[q] += by; pb1[q] += by; paa0[q] += by; paa1[q] += by; phc[q] += by; pha[q] += by; pruu[q] += by; }
        };
        auto stage = [&](auto sym_tag, auto off_tag) {
            constexpr bool SYM = decltype(sym_tag)::value;
            constexpr int OFF = decltype(off_tag)::value * RC::SIZE;          // the record to prefetch, from the operand addresses
            wdec += w_inc;
            pdec += p_inc;
            double nb0[NI], nb1[NI], naa0[NI], naa1[NI], nhc[NI], nha[NI], nruu0[NI], nruu1[NI];
            double L00[NI], L01[NI], L11[NI], det[NI], rc[NI], er[NI], cl[NI], S0[NI], S1[NI], B0[NI], B1[NI], T[NI], G[NI], Gs[NI], Kt[NI], Ke[NI];
            // The wavefront issues in order: the scalar chain Lam -> det -> 1 / det (ten dependent instructions) is cut into pieces that
            // are laid between the steps of the matrix chain, and the scheduling fences keep the compiler from clumping them again.
#define MPC_FENCE() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
            for (int q = 0; q < NI; ++q) {        // operands of the next stage (software pipeline; stage -1 reads the pad record)
                nb0[q] = pb0[q][OFF]; nb1[q] = pb1[q][OFF]; naa0[q] = paa0[q][OFF]; naa1[q] = paa1[q][OFF]; nhc[q] = phc[q][OFF]; nha[q] = pha[q][OFF];
                nruu0[q] = pruu[q][OFF]; nruu1[q] = pruu[q][OFF + 1];
            }
#pragma unroll
            for (int q = 0; q < NI; ++q) {        // Y = M+ At, one instruction per column block; Lam = Ruu + B'P+B (+ delta_w)
                const double P22 = wv_readlane(M[q], 34), P23 = wv_readlane(M[q], 35), P33 = wv_readlane(M[q], 51);
                S0[q] = wv_mfma(M[q], b0[q], 0.0);
                S1[q] = wv_mfma(M[q], b1[q], 0.0);
                L00[q] = ruu0[q] + dt2 * P22; L01[q] = dt2 * P23; L11[q] = ruu1[q] + dt2 * P33;
                if (SYM) { L00[q] += delta[q]; L11[q] += delta[q]; }               // (the other variant runs with delta_w = 0 on every instance)
                cl[q] = (x == 0) ? L11[q] : L00[q];                                 // the diagonal entry of adj(Lam) this lane's row of Gt meets
            }
            MPC_FENCE();
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                // (SYM -- instances with heavily weighted circle rows: Lam = Ruu + w g g' is dominated by a rank-one term, L00 L11 and L01^2 agree
                //  to the first ten digits; Kahan's difference of products keeps the rounding error of L01^2, which is as large as det itself)
                const double pp = L01[q] * L01[q];
                det[q] = fma(L00[q], L11[q], -pp);
                if (SYM) det[q] -= symq[q] ? fma(L01[q], L01[q], -pp) : 0.0;            // (per instance: an instance's bits do not depend on its wavefront mate)
            }
            MPC_FENCE();
#pragma unroll
            for (int q = 0; q < NI; ++q) { rc[q] = __builtin_amdgcn_rcp(det[q]); S0[q] = wv_sum_hi(S0[q]); }       // S0: lane (x, *, lo, y) = Y[4 lo + x][y]
            MPC_FENCE();
#pragma unroll
            for (int q = 0; q < NI; ++q) { er[q] = fma(-det[q], rc[q], 1.0); S1[q] = wv_sum_hi(S1[q]); }           // S1:                    = Y[4 lo + x][4 + y]
            MPC_FENCE();
#pragma unroll
            for (int q = 0; q < NI; ++q) { rc[q] = fma(er[q], rc[q], rc[q]); B0[q] = wv_dpp<DPP_SHR4, 0xA>(S0[q], S1[q]); }   // k-step 0: block (hi, lo) = Y_{0, lo}
            MPC_FENCE();
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                er[q] = fma(-det[q], rc[q], 1.0);
                B1[q] = wv_dpp<DPP_SHL4, 0x5>(S1[q], S0[q]);                       // k-step 1: block (hi, lo) = Y_{1, lo}
                T[q] = wv_mfma(aa0[q], B0[q], hc[q]);                              // T = At' Y + Ht
            }
            MPC_FENCE();
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                rc[q] = fma(er[q], rc[q], rc[q]);
                // Gt = dt * rows (2, 3) of Y + [Hux | gu] in rows x = 0, 1 (both halves)
                G[q] = dt * wv_upper_to_lower(B0[q]) + ha[q];
                T[q] = wv_mfma(aa1[q], B1[q], T[q]);
            }
            MPC_FENCE();
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                Gs[q] = wv_swap16(G[q]);
                good[q] = good[q] && (L00[q] > 0.0) && (det[q] > 0.0);
                Kt[q] = -cl[q] * G[q];                                              // -adj(Lam) Gt = L01 Gs - cl G, without waiting for 1 / det
                if (SYM) Ke[q] = symq[q] ? fma(-cl[q], G[q], -Kt[q]) : 0.0;                         // (the rounding error of that product: the two terms cancel like those of det)
            }
            MPC_FENCE();
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const double GA = wv_dpp<DPP_SHL4, 0x6>(G[q], G[q]);               // block (hi, lo) = column block hi of Gt (read transposed)
                Kt[q] = fma(L01[q], Gs[q], Kt[q]);
                if (SYM) Kt[q] += Ke[q];
                Kt[q] *= rc[q];                                                     // Kt = -Lam^-1 Gt  (rc = 1 / det: v_rcp_f64 + two Newton steps)
                M[q] = wv_mfma(GA, Kt[q], T[q]);                                   // M = T + Gt' Kt + delta_w I
                if (SYM) M[q] += delta[q] * m.dmask;
                // (per instance, so that the result of an instance does not depend on which instance shares its wavefront)
                if (SYM) { const double Ms = 0.5 * (M[q] + wv_bpermute(M[q], m.tr_addr)); M[q] = symq[q] ? Ms : M[q]; }
                rec[q][w_first[q] - wdec] = Kt[q];                                              // gains for the forward sweep (lanes without an entry: the dump area)
                rec[q][p_first[q] - pdec] = M[q];                                               // cost-to-go for the stage threads, over this stage's consumed Ruu / gu / gx / H
                b0[q] = nb0[q]; b1[q] = nb1[q]; aa0[q] = naa0[q]; aa1[q] = naa1[q]; hc[q] = nhc[q]; ha[q] = nha[q]; ruu0[q] = nruu0[q]; ruu1[q] = nruu1[q];
            }
#undef MPC_FENCE
        };
        auto alive = [&]() {
            bool a = false;
#pragma unroll
            for (int q = 0; q < NI; ++q) a = a || (need[q] && good[q]);
            return __builtin_amdgcn_ballot_w64(a) != 0ull;
        };
        // (a sweep whose every instance has met an indefinite Lam is abandoned: it is repeated with a larger delta_w anyway)
        // (two stages per trip: the operand registers of the software pipeline swap roles without copies)
        using Off0 = std::integral_constant<int, 0>;
        using Off1 = std::integral_constant<int, 1>;
        if (__builtin_amdgcn_ballot_w64(symm) != 0ull) {
            int k = N - 1;
            for (; k >= 1 && alive(); k -= 2) { stage(std::true_type{}, Off1{}); stage(std::true_type{}, Off0{}); bump(-2 * RC::SIZE); }
            if (k == 0 && alive()) { bump(RC::SIZE); stage(std::true_type{}, Off0{}); }
        } else {
            int k = N - 1;
            for (; k >= 1 && alive(); k -= 2) { stage(std::false_type{}, Off1{}); stage(std::false_type{}, Off0{}); bump(-2 * RC::SIZE); }
            if (k == 0 && alive()) { bump(RC::SIZE); stage(std::false_type{}, Off0{}); }
        }
        bool again = false;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            if (!need[q]) continue;
            if (good[q]) { need[q] = false; ok[q] = true; continue; }
            // IPOPT's inertia-correction schedule (Waechter & Biegler section 3.1), as riccati_instance
            if (delta[q] == 0.0) delta[q] = (in[q].delta_last == 0.0) ? DW_0 : fmax(DW_MIN, KW_MINUS * in[q].delta_last);
            else delta[q] *= (in[q].delta_last == 0.0) ? KW_PLUS_BAR : KW_PLUS;
            if (delta[q] > DW_MAX) need[q] = false; else again = true;
        }
        if (__builtin_amdgcn_ballot_w64(again) == 0ull) return;
    }
}

// ---- forward sweeps of NI instances: dz_k = (du_k, dx_k) into the records (Rec::DU / Rec::DX: over A[0..1] and -c of the stage, which the
// sweep has consumed by the time it writes them) ------------------------------------------------------------------------------------------
// One matrix instruction per stage: x~' = Acl' x~ with Acl' = At + Bt Kt in the state rows and the two rows of Kt in rows 6, 7 (the
// affine coordinate is put back by hand), the k index spread over the hardware blocks (block (hi, lo) = Acl'_{lo, hi} x~_hi) and summed over
// hi by a DPP row rotation.  x0[q]: B-operand register, lane (x, hi, *, 0) = x~_0[4 hi + x] (the affine coordinate 7 holds 1), zero for y != 0.
// Lanes without an entry to store write the dump area.
template <int NX, int NI>
__device__ __forceinline__ void mfma_forward(const PRef& P, const MfmaLane<NX>& m, const MfmaInst (&in)[NI], const mpc_lds_ptr (&rec)[NI], int lane,
                                             mpc_lds_ptr dump, const double (&x0)[NI], const bool (&ok)[NI]) {
    using RC = Rec<NX>;
    const int N = P.N;
    const bool st = m.dz_row >= 0;
    // per-lane running offset from the records of the instance: du lanes write DU + row of record k, dx lanes DX + state of record k + 1
    const int inc = st ? RC::SIZE : 0;
    const double fixval = (m.fix == 2) ? 1.0 : 0.0;
    const bool fix = m.fix != 0;
    double X[NI], f1[NI], f2[NI];
    int doff[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        X[q] = x0[q];
        doff[q] = st ? (m.dz_next ? RC::SIZE + RC::DX + (m.dz_row - 2) : RC::DU + m.dz_row) : (int)(dump - rec[q]) + lane;
        mpc_lds_cptr r = rec[q];
        f1[q] = r[m.oF1]; f2[q] = r[m.oF2];              // (before dx_0 goes over the -c entries of record 0)
        // dx_0: the storing lanes (x, hi = 0, lo, 0) want row 4 lo + x, the operand layout carries row 4 hi + x: one bank move
        const double xs = wv_dpp<DPP_SHR4, 0x2>(X[q], wv_dpp<DPP_SHL8, 0x1>(X[q], X[q]));      // bank (0, 1) <- bank (1, *)
        if (ok[q] && m.dz_next) rec[q][doff[q] - RC::SIZE] = xs;
    }
    // The result of a stage is stored one stage LATER, behind the operand reads of the stage after: the LDS returns in order, so a write
    // between the product and the next reads would sit in front of them in the queue the chain waits on (177 -> 141 ticks per stage).
    double Sprev[NI];
    auto flush = [&]() {
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            if (ok[q]) rec[q][doff[q]] = Sprev[q];
            doff[q] += inc;
        }
    };
    auto fstage = [&](int k, auto first_tag) {
        double n1[NI], n2[NI], S[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            mpc_lds_cptr rn = rec[q] + (k + 1) * RC::SIZE;                                  // (stage N: the terminal record, read and not used)
            n1[q] = rn[m.oF1]; n2[q] = rn[m.oF2];
        }
        // (dx_k over record k's -c, du_{k-1} over record k - 1's A: both records' operands were read before -- LDS operations of a wavefront keep their order)
        if (!decltype(first_tag)::value) flush();
#pragma unroll
        for (int q = 0; q < NI; ++q) S[q] = wv_mfma(f1[q] + m.fscale * f2[q], X[q], 0.0);
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            S[q] = wv_sum_hi(S[q]);                                              // lane (x, *, lo, 0) = x~'[4 lo + x]: dx_{k+1}, du_k in rows 6, 7
            Sprev[q] = S[q];
            S[q] = fix ? fixval : S[q];
            X[q] = wv_dpp<DPP_SHL4, 0x6>(S[q], S[q]);                            // block (hi, lo) <- x~_hi
            f1[q] = n1[q]; f2[q] = n2[q];
        }
    };
    if (N >= 1) {
        fstage(0, std::true_type{});
        int k = 1;
        for (; k + 1 < N; k += 2) { fstage(k, std::false_type{}); fstage(k + 1, std::false_type{}); }
        if (k < N) fstage(k, std::false_type{});
        flush();
    }
    if (st && !m.dz_next) {                                                  // du_N = 0
#pragma unroll
        for (int q = 0; q < NI; ++q)
            if (ok[q]) rec[q][doff[q]] = 0.0;
    }
}

#endif   // __HIP_DEVICE_COMPILE__

}  // namespace mpc
