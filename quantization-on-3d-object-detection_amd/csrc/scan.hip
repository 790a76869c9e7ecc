// Exclusive prefix sum over an int32 flag array whose live length sits in device memory.
// Three launches (chunk sums -> scan of chunk sums -> chunk scan); 2048 items per 256-thread block,
// wave-level shuffles + one LDS hop. Used by the voxelizer and the strided-conv output generator to
// turn "first candidate of its cell" flags into dense, deterministic row ids.
#include "common.h"
#include "scan.h"

#define SCAN_BLOCK 256
#define SCAN_ITEMS 8
#define SCAN_CHUNK (SCAN_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ int live_len(const int *count, int mult, int n_max) {
    long long n = (long long)(*count) * mult;
    if (n > n_max) n = n_max;
    if (n < 0) n = 0;
    return (int)n;
}

// inclusive scan across the block of one int per thread; returns exclusive prefix, total in *block_total
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds_wave_sums, int *block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) lds_wave_sums[wave] = incl;
    __syncthreads();
    int wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        int s = lds_wave_sums[w];
        if (w < wave) wave_base += s;
        total += s;
    }
    __syncthreads();
    *block_total = total;
    return wave_base + incl - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_chunk_sums(const int *__restrict__ flags, const int *count,
                                                              int mult, int n_max, int *__restrict__ chunk_sums) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    const int n = live_len(count, mult, n_max);
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        int i = base + j;
        if (i < n) s += flags[i];
    }
    int total;
    (void)block_exclusive_scan(s, wsum, &total);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void scan_of_chunk_sums(int *__restrict__ chunk_sums, int nchunks, int *total_out) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nchunks; base += 1024) {
        int i = base + threadIdx.x;
        int v = (i < nchunks) ? chunk_sums[i] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wave_base = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            int s = wsum[w];
            if (w < wave) wave_base += s;
            total += s;
        }
        int carry = carry_s;
        if (i < nchunks) chunk_sums[i] = carry + wave_base + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_chunks(const int *__restrict__ flags, const int *count, int mult,
                                                          int n_max, const int *__restrict__ chunk_sums,
                                                          int *__restrict__ prefix) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    const int n = live_len(count, mult, n_max);
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    if (blockIdx.x * SCAN_CHUNK >= n) return;  // whole block exits together
    int f[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        int i = base + j;
        f[j] = (i < n) ? flags[i] : 0;
        s += f[j];
    }
    int total;
    int excl = block_exclusive_scan(s, wsum, &total) + chunk_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        int i = base + j;
        if (i < n) prefix[i] = excl;
        excl += f[j];
    }
}

int aq3d_scan_chunks_for(int n_max) { return aq3d_div_up(n_max > 0 ? n_max : 1, SCAN_CHUNK); }

int aq3d_scan_flags(const int *flags, const int *count, int mult, int n_max, int *prefix, int *total,
                    int *chunk_sums, hipStream_t stream) {
    const int nchunks = aq3d_scan_chunks_for(n_max);
    hipLaunchKernelGGL(scan_chunk_sums, dim3(nchunks), dim3(SCAN_BLOCK), 0, stream, flags, count, mult, n_max, chunk_sums);
    hipLaunchKernelGGL(scan_of_chunk_sums, dim3(1), dim3(1024), 0, stream, chunk_sums, nchunks, total);
    hipLaunchKernelGGL(scan_chunks, dim3(nchunks), dim3(SCAN_BLOCK), 0, stream, flags, count, mult, n_max, chunk_sums, prefix);
    AQ3D_CHECK_LAUNCH("aq3d_scan_flags");
    return AQ3D_OK;
}
