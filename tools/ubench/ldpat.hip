// Micro-benchmark of the stage kernel's load pattern: 256-thread workgroups (4 waves), thread (k, bl) with bl = t % 8,
// k = t / 8, reads NROW rows of a tile-major workspace [tile][row][64 lanes] (8 B).  Variants:
//   0: one buffer_load_dwordx2 per (row) per thread  -- 8 x 64-byte segments per wave instruction (current kernel)
//   1: global->LDS DMA, 16 B per lane: 16 segments per wave instruction, then ds_read_b64 per row
//   2: rows interleaved in pairs ([row/2][lane][2]): one dwordx4 per row pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int NROW = 143, NK = 31, R = NROW;           // rows per stage (all arrays lumped), stages
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}

template <int VAR>
__global__ void __launch_bounds__(256) k_ld(const double* ws, unsigned ws_bytes, unsigned tile_bytes, double* out, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, bl = t & 7, k = t >> 3;
    unsigned blk = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
    const unsigned tile = blk >> 3, b0 = (blk & 7u) * 8u;
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(ws, ws_bytes);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double acc = 0.0;
    if (VAR == 0) {
        if (k < NK) {
            const unsigned voff = tile * tile_bytes + (unsigned)k * (R * 512u) + (b0 + bl) * 8u;
            double v[NROW];
#pragma unroll
            for (int r = 0; r < NROW; ++r) v[r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(voff + r * 512u), 0, 0));
#pragma unroll
            for (int r = 0; r < NROW; ++r) acc += v[r];
        }
    } else if (VAR == 1) {
        if (k < NK) {
            const unsigned voff = tile * tile_bytes + (unsigned)k * (R * 512u) + (b0 + bl) * 8u;
#pragma unroll
            for (int r = 0; r < NROW; ++r) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, (double)(r + t)), rs, (int)(voff + r * 512u), 0, 0);
        }
    } else if (VAR == 3) {
        if (k < NK) {
            const unsigned voff = tile * tile_bytes + (unsigned)k * (R * 512u) + (b0 + bl) * 16u;
#pragma unroll
            for (int r = 0; r < NROW; r += 2) {
                const v2u a = __builtin_bit_cast(v2u, (double)(r + t)), b = __builtin_bit_cast(v2u, (double)(r - t));
                __builtin_amdgcn_raw_buffer_store_b128(v4u{a.x, a.y, b.x, b.y}, rs, (int)(voff + (r >> 1) * 1024u), 0, 0);
            }
        }
    } else {
        if (k < NK) {
            // pair-interleaved layout: element (row, lane) at ((row >> 1) * 128 + lane * 2 + (row & 1)) * 8 bytes within the stage
            const unsigned voff = tile * tile_bytes + (unsigned)k * (R * 512u) + (b0 + bl) * 16u;
            double v[NROW + 1];
#pragma unroll
            for (int r = 0; r < NROW; r += 2) {
                const v4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(voff + (r >> 1) * 1024u), 0, 0);
                v[r] = __builtin_bit_cast(double, v2u{q.x, q.y});
                v[r + 1] = __builtin_bit_cast(double, v2u{q.z, q.w});
            }
#pragma unroll
            for (int r = 0; r < NROW; ++r) acc += v[r];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * 256 + t] = acc;
    if (t == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    const int ntiles = 64, nblk = ntiles * 8;
    const unsigned tile_bytes = (unsigned)(NROW + 1) * NK * 512u;
    const size_t bytes = (size_t)ntiles * tile_bytes;
    double *ws, *out; unsigned long long* clk;
    CK(hipMalloc(&ws, bytes)); CK(hipMalloc(&out, (size_t)nblk * 256 * 8)); CK(hipMalloc(&clk, nblk * 8));
    CK(hipMemset(ws, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = (size_t)((NROW * NK + 15) / 16) * 1024;
    for (int var = 0; var < 4; ++var) {
        float best = 1e9f; double cyc = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            if (var == 0) hipLaunchKernelGGL(k_ld<0>, dim3(nblk), dim3(256), 0, 0, ws, (unsigned)bytes, tile_bytes, out, clk);
            if (var == 1) hipLaunchKernelGGL(k_ld<1>, dim3(nblk), dim3(256), 0, 0, ws, (unsigned)bytes, tile_bytes, out, clk);
            if (var == 3) hipLaunchKernelGGL(k_ld<3>, dim3(nblk), dim3(256), 0, 0, ws, (unsigned)bytes, tile_bytes, out, clk);
            if (var == 2) hipLaunchKernelGGL(k_ld<2>, dim3(nblk), dim3(256), 0, 0, ws, (unsigned)bytes, tile_bytes, out, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<unsigned long long> h(nblk);
        CK(hipMemcpy(h.data(), clk, nblk * 8, hipMemcpyDeviceToHost));
        for (auto c : h) cyc += (double)c;
        printf("variant %d (0: 8B loads, 1: 8B stores, 2: 16B pair loads, 3: 16B pair stores): %.1f us per launch (%d WGs, %.1f MB), mean %.0f ticks per WG, LDS %zu\n", var, best * 1e3, nblk, bytes / 1e6, cyc / nblk, var == 1 ? lds : 0);
    }
    return 0;
}
