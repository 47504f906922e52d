// Stand-alone reproducer of the stale read behind round 5's "paired-store corruption" (profiles/r05_store_pairing.txt) and the measurement
// behind the hand-off protocol of k_pipeline (csrc/mpcgpu.hip, DevParams::xcu): what does a compute unit read from a 128-byte line that
// ANOTHER compute unit of its XCD has rewritten since it last touched it?
//
// Two workgroups of one wavefront on one XCD (picked by HW_REG_XCC_ID from a grid of 64; different CUs by HW_REG_HW_ID), the consumer C and
// the producer P, ping-pong over agent-scope flags.  One trial:
//   C: touches line L the way the variant says (loads it / stores it whole / stores it in 8-byte pieces)          -> L may now sit in C's vector L1
//   C: optionally streams `evict_kb` KB of other lines through its L1 (stride given: contiguous, or the 1 KiB row-pair stride of a stage item)
//   P: rewrites L with the trial's new value (plain stores), s_waitcnt vmcnt(0), raises the flag                 -> the new value is in the XCD's L2
//   C: sees the flag (agent-scope atomic load), applies the variant's "acquire", loads L again, counts lanes that still see the OLD value
// Variants of the acquire:  none | buffer_inv sc0 (+ workgroup fence) | agent-scope acquire fence (buffer_inv sc1) | none, but the re-load carries sc1
// Variants of C's first touch: 16-byte loads (8 lanes = the whole line) | whole-line store (8 lanes x 16 bytes: what the paired (mu, tau) store was) |
//                              sixteen 8-byte stores (what the product's scalar rows are) | whole-line store followed by one 8-byte store into it
// Output: stale lanes / lanes read, per combination.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 stale_l1.hip -o stale_l1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

enum Touch { T_LOAD16 = 0, T_STORE_LINE = 1, T_STORE_8B = 2, T_STORE_LINE_THEN_8B = 3, T_COUNT };
enum Acq { A_NONE = 0, A_INV_SC0 = 1, A_AGENT = 2, A_SC1_LOAD = 3, A_COUNT };

struct Args {
    unsigned char* buf;          // line L at buf + line_off; the eviction stream behind it
    unsigned int* ctl;           // [0] C's id, [1] P's id, [2] flag C -> P, [3] flag P -> C, [4] stale lanes, [5] lanes read, [6] roles taken, [8..] scratch
    unsigned int xcc;            // the XCD both run on
    unsigned int trials, touch, acq, evict_kb, evict_stride, line_off;
};

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// bounded wait (100 MHz wall clock: 0.2 s): a partner that never came must not hang the box
__device__ __forceinline__ bool wait_ge(unsigned* ctl, int word, unsigned v) {
    const unsigned long long t0 = wall_clock64();
    while (ld_flag(ctl + word) < v) {
        if (ld_flag(ctl + 10) != 0u) return false;
        if (wall_clock64() - t0 > 20000000ull) { st_flag(ctl + 10, 1u); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

__global__ void __launch_bounds__(64) k_stale(const Args A) {
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;       // HW_REG_XCC_ID[3:0]
    if (xcc != A.xcc) return;
    __shared__ unsigned role_s;
    const unsigned hw_id = (unsigned)__builtin_amdgcn_s_getreg((15 << 11) | 4) | 0x80000000u;      // HW_REG_HW_ID[15:0]: wave, simd, pipe, cu [11:8], sh [12], se [15:13]
    if (threadIdx.x == 0) {
        // first arrival on the XCD: consumer; the first later arrival that sits on ANOTHER compute unit: producer; the rest leave
        unsigned role = 2u;
        if (atomicAdd(A.ctl + 6, 1u) == 0u) { role = 0u; st_flag(A.ctl + 0, hw_id); }
        else {
            unsigned c_id;
            const unsigned long long t0 = wall_clock64();
            while ((c_id = ld_flag(A.ctl + 0)) == 0u && wall_clock64() - t0 < 20000000ull) __builtin_amdgcn_s_sleep(1);
            if (((c_id ^ hw_id) & 0xFF00u) != 0u && atomicCAS(A.ctl + 1, 0u, hw_id) == 0u) role = 1u;
        }
        role_s = role;
    }
    __syncthreads();
    const unsigned role = role_s;
    if (role > 1u) return;
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(A.buf, 0, 0x7FFFFFFF, 0x00020000);
    const int off16 = (int)A.line_off + lane * 16;            // lanes 0..7 cover the line with 16-byte pieces
    unsigned stale = 0, seen = 0, sink = 0;
    for (unsigned i = 1; i <= A.trials; ++i) {
        const unsigned vold = 2 * i, vnew = 2 * i + 1;
        if (role == 0u) {
            // ---- consumer: first touch (the line holds vold: the producer of the previous trial wrote 2 (i - 1) + 1, so write vold first -- through P)
            //      protocol: P writes vold (phase a), C touches, P writes vnew (phase b), C re-reads
            if (!wait_ge(A.ctl, 3, 2 * i - 1)) break;                                      // P has written vold
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                              // (a real acquire here: the first touch must see vold)
            if (A.touch == T_LOAD16) {
                if (lane < 8) { const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, off16, 0, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); sink += v.x; }
            } else if (A.touch == T_STORE_LINE || A.touch == T_STORE_LINE_THEN_8B) {
                if (lane < 8) __builtin_amdgcn_raw_buffer_store_b128(v4u{vold, vold, vold, vold}, rs, off16, 0, 0);
                if (A.touch == T_STORE_LINE_THEN_8B && lane < 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_raw_buffer_store_b64(v2u{vold, vold}, rs, off16, 0, 0); }
            } else {
                if (lane < 16) __builtin_amdgcn_raw_buffer_store_b64(v2u{vold, vold}, rs, (int)A.line_off + lane * 8, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ---- optional: other lines through the L1 (a stage item loads ~130 KB at its top)
            if (A.evict_kb) {
                const unsigned n = A.evict_kb * 1024u / (64u * 16u);                        // wave instructions of 1 KiB
                v4u acc = {0, 0, 0, 0};
                for (unsigned q = 0; q < n; ++q) {
                    // contiguous: 64 lanes x 16 bytes in a row; row-pair stride: 8 lanes x 16 bytes = one line per KiB, like the 8 instance columns of a stage item
                    const int o = A.evict_stride == 0 ? (int)(4096u + q * 1024u) + lane * 16 : (int)(4096u + (q * 8u + (unsigned)(lane >> 3)) * 1024u + A.line_off % 1024u) + (lane & 7) * 16;
                    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
                    acc.x += v.x;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (acc.x == 0xDEADBEEFu) A.ctl[8] = acc.x;
            }
            if (lane == 0) st_flag(A.ctl + 2, 2 * i);                                        // "touched": P may rewrite
            if (!wait_ge(A.ctl, 3, 2 * i)) break;                                          // P has written vnew
            // ---- the acquire under test
            if (A.acq == A_INV_SC0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); asm volatile("buffer_inv sc0" ::: "memory"); }
            else if (A.acq == A_AGENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane < 8) {
                const v4u v = A.acq == A_SC1_LOAD ? __builtin_amdgcn_raw_buffer_load_b128(rs, off16, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(rs, off16, 0, 0);
                seen += 1;
                stale += (v.x == vold) ? 1u : 0u;
                if (v.x != vold && v.x != vnew) A.ctl[9] = v.x;                             // (neither: a torn / foreign value -- never seen)
            }
        } else {
            // ---- producer
            if (!wait_ge(A.ctl, 7, i - 1)) break;                                          // C has finished re-reading trial i - 1
            if (lane < 8) __builtin_amdgcn_raw_buffer_store_b128(v4u{vold, vold, vold, vold}, rs, off16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_flag(A.ctl + 3, 2 * i - 1);
            if (!wait_ge(A.ctl, 2, 2 * i)) break;                                          // C has touched
            if (lane < 8) __builtin_amdgcn_raw_buffer_store_b128(v4u{vnew, vnew, vnew, vnew}, rs, off16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               // acknowledged by the L2 (plain stores: the pipeline's producer side)
            if (lane == 0) st_flag(A.ctl + 3, 2 * i);
        }
        if (role == 0u) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) st_flag(A.ctl + 7, i); }
    }
    if (role == 0u) {
        for (int m = 1; m < 8; m <<= 1) { stale += __shfl_xor(stale, m, 64); seen += __shfl_xor(seen, m, 64); }
        if (lane == 0) { A.ctl[4] = stale; A.ctl[5] = seen; if (sink == 0xDEADBEEFu) A.ctl[8] = sink; }
    }
}

int main(int argc, char** argv) {
    const unsigned trials = argc > 1 ? (unsigned)atoi(argv[1]) : 2000u;
    unsigned char* buf; unsigned* ctl;
    const size_t bytes = 64u << 20;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&ctl, 64 * sizeof(unsigned)) != hipSuccess) { printf("no device memory\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    static const char* tn[T_COUNT] = {"16-byte loads of the line", "whole-line store (8 x 16 B)", "sixteen 8-byte stores", "whole-line store + one 8-byte store"};
    static const char* an[A_COUNT] = {"no acquire", "buffer_inv sc0 + workgroup fence", "agent-scope acquire (buffer_inv sc1)", "no acquire, re-load with sc1"};
    auto run = [&](unsigned touch, unsigned acq, unsigned evict_kb, unsigned stride, unsigned line_off, bool print = true) {
        (void)hipMemset(ctl, 0, 64 * sizeof(unsigned));
        Args A{buf, ctl, 0u, trials, touch, acq, evict_kb, stride, line_off};
        hipLaunchKernelGGL(k_stale, dim3(64), dim3(64), 0, 0, A);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
        unsigned h[16]; (void)hipMemcpy(h, ctl, sizeof h, hipMemcpyDeviceToHost);
        if (print) printf("  %-38s | %-38s | evict %4u KB %-10s | line +%5u : stale %6u of %6u lanes (%.1f %%)%s%s\n", tn[touch], an[acq], evict_kb,
                          evict_kb ? (stride ? "(1 KiB stride)" : "(contiguous)") : "", line_off, h[4], h[5], h[5] ? 100.0 * h[4] / h[5] : 0.0,
                          (h[1] == 0u || h[10] != 0u) ? "  [no producer on another CU / a wait ran out: nothing measured]" : "",
                          h[9] ? "  [foreign value seen]" : "");
        return h[4];
    };
    printf("consumer and producer: two single-wavefront workgroups on XCD 0, %u trials per line\n", trials);
    printf("1. what the consumer's acquire must be (no other traffic through its L1):\n");
    for (unsigned touch = 0; touch < T_COUNT; ++touch)
        for (unsigned acq = 0; acq < A_COUNT; ++acq) run(touch, acq, 0, 0, 0);
    printf("2. does other traffic push the line out?  (whole-line store, no acquire; a stage item of k_pipeline loads ~130 KB at its top)\n");
    for (unsigned kb : {16u, 32u, 64u, 128u, 256u})
        for (unsigned stride : {0u, 1u}) run(T_STORE_LINE, A_NONE, kb, stride, 0);
    printf("3. the two scalar row pairs of round 5's experiment: (mu, tau) = row pair 0, (fcost, logsum) = row pair 2 (+ 2048 bytes) of the per-instance\n"
           "   scalar array, behind a stage item's loads (1 KiB stride, the line's own column) -- whole-line store, no acquire:\n");
    for (unsigned kb : {0u, 32u, 128u})
        for (unsigned off : {0u, 2048u}) run(T_STORE_LINE, A_NONE, kb, 1, off);
    return 0;
}
