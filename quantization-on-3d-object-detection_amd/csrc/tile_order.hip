// Tile order of a gather table: rows with the same neighbour pattern next to each other.
//
// The output-stationary sparse conv processes 128 (or 16) output rows per tile and skips a kernel offset only when NO row of the
// tile has a neighbour there. In voxelizer order (points are shuffled, data_processor.py:95-105) a 128-row tile touches all 27
// offsets although a row has 2-14 neighbours: the matrix pipe multiplies zeros. spconv 2.x meets the same problem with its
// "mask argsort"; here every chunk of 16 384 rows is sorted by its 27-bit occupancy mask in LDS (one workgroup per chunk, bitonic
// network over (mask << 32 | row) words), which keeps rows of equal pattern together: measured on the W150k levels the executed
// (tile, offset) units drop to 0.29 / 0.54 / 0.63 / 0.75 of rows x 27 for the four submanifold tables (a global sort: 0.25 /
// 0.53 / 0.60 / 0.72) and to 0.19 - 0.44 for the strided ones. The conv kernels read the table in tile order and write row
// perm[i]; sums per output row are unchanged (the same offsets in the same order, zeros skipped), so results are bit-identical
// to the identity order.
#include "common.h"

typedef unsigned long long u64;
#define TO_CHUNK 16384
#define TO_THREADS 1024

__global__ __launch_bounds__(TO_THREADS) void tile_order_kernel(const int *__restrict__ nbr, const int *count, int cap, int kvol,
                                                                int *__restrict__ perm, int *__restrict__ nbr_sorted) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *s = reinterpret_cast<u64 *>(smem_raw);                     // [TO_CHUNK]
    int n = *count;
    if (n > cap) n = cap;
    const int base = (int)blockIdx.x * TO_CHUNK;
    if (base >= n) return;
    const int rows = n - base < TO_CHUNK ? n - base : TO_CHUNK;
    // sort size: the power of two covering this chunk's live rows (a short last chunk sorts less)
    int N = 128;
    while (N < rows) N <<= 1;
    for (int r = threadIdx.x; r < N; r += TO_THREADS) {
        u64 key = ~0ull;                                              // padding sorts last
        if (r < rows) {
            unsigned m = 0;
            const int *row = nbr + (size_t)(base + r) * kvol;
            for (int k = 0; k < kvol; ++k) m |= (row[k] >= 0 ? 1u : 0u) << k;
            key = ((u64)m << 32) | (unsigned)r;
        }
        s[r] = key;
    }
    __syncthreads();
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (N >> 1); t += TO_THREADS) {
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                const u64 a = s[i], b = s[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { s[i] = b; s[l] = a; }
            }
            __syncthreads();
        }
    for (int r = threadIdx.x; r < rows; r += TO_THREADS) perm[base + r] = base + (int)(unsigned)(s[r] & 0xffffffffu);
    // table in tile order: kvol consecutive threads copy one row (coalesced on the write side)
    const int total = rows * kvol;
    for (int t = threadIdx.x; t < total; t += TO_THREADS) {
        const int r = t / kvol, k = t - r * kvol;
        const int src = base + (int)(unsigned)(s[r] & 0xffffffffu);
        nbr_sorted[(size_t)(base + r) * kvol + k] = nbr[(size_t)src * kvol + k];
    }
}

/* include/aq3d.h: aq3d_tile_order */
extern "C" int aq3d_tile_order(const int32_t *nbr, const int32_t *count, int32_t cap, int32_t kvol, int32_t *perm_out,
                               int32_t *nbr_sorted_out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(nbr && count && perm_out && nbr_sorted_out, "aq3d_tile_order: null argument");
    AQ3D_REQUIRE(cap >= 1 && kvol >= 1 && kvol <= 27, "aq3d_tile_order: cap=%d kvol=%d", cap, kvol);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(tile_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TO_CHUNK * 8);
        attr = true;
    }
    hipLaunchKernelGGL(tile_order_kernel, dim3(aq3d_div_up(cap, TO_CHUNK)), dim3(TO_THREADS), TO_CHUNK * 8, stream, nbr, count, cap, kvol,
                       perm_out, nbr_sorted_out);
    AQ3D_CHECK_LAUNCH("aq3d_tile_order");
    return AQ3D_OK;
}
