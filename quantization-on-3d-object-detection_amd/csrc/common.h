// Shared helpers of the gfx950 kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/aq3d.h"

#define AQ3D_WAVE 64

void aq3d_set_error(const char *fmt, ...);

#define AQ3D_REQUIRE(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            aq3d_set_error(__VA_ARGS__);         \
            return AQ3D_ERR_INVALID_ARG;         \
        }                                        \
    } while (0)

#define AQ3D_CHECK_LAUNCH(name)                                                    \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            aq3d_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return AQ3D_ERR_LAUNCH;                                                \
        }                                                                          \
    } while (0)

static inline int aq3d_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int aq3d_is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// Grid for a grid-stride loop over `work` items with `block` threads: enough blocks to fill
// 256 CUs x 8 blocks, never more than the work needs.
// Zero fill as an ordinary kernel of the caller's stream (16-byte stores, grid-stride): a hipMemsetAsync becomes a MEMSET NODE of a
// captured hipGraph, which this runtime does not run like the kernel nodes around it (scratch/ablate.py, round 4).
__global__ __launch_bounds__(256) static void aq3d_zero_kernel(uint4 *__restrict__ p, size_t n16, int tail) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) reinterpret_cast<unsigned char *>(p + n16)[threadIdx.x] = 0;
}
// p 16-byte aligned (device allocations are); any byte count
static inline void aq3d_zero_async(void *p, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return;
    const size_t n16 = bytes / 16;
    const size_t blocks = n16 ? (n16 + 255) / 256 : 1;
    hipLaunchKernelGGL(aq3d_zero_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), 0, stream, (uint4 *)p, n16, (int)(bytes % 16));
}

static inline int aq3d_grid(int64_t work, int block, int max_blocks = 2048) {
    int64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

#ifdef __HIPCC__
// ---- coordinate hash (open addressing, linear probing, 64-bit keys, empty = -1) ----
__device__ __forceinline__ uint32_t aq3d_hash64(int64_t key) {
    uint64_t k = (uint64_t)key;
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return (uint32_t)k;
}

// Returns the slot of `key`, inserting it if absent. Table must not be full.
__device__ __forceinline__ int aq3d_hash_insert(int64_t *keys, int table_size, int64_t key) {
    uint32_t mask = (uint32_t)table_size - 1u;
    uint32_t slot = aq3d_hash64(key) & mask;
    for (int probe = 0; probe < table_size; ++probe) {
        unsigned long long prev = atomicCAS((unsigned long long *)&keys[slot], (unsigned long long)(-1LL),
                                            (unsigned long long)key);
        if (prev == (unsigned long long)(-1LL) || prev == (unsigned long long)key) return (int)slot;
        slot = (slot + 1u) & mask;
    }
    return -1;
}

// Returns the slot of `key` or -1. Read-only (table complete).
__device__ __forceinline__ int aq3d_hash_find(const int64_t *keys, int table_size, int64_t key) {
    uint32_t mask = (uint32_t)table_size - 1u;
    uint32_t slot = aq3d_hash64(key) & mask;
    for (int probe = 0; probe < table_size; ++probe) {
        int64_t k = keys[slot];
        if (k == key) return (int)slot;
        if (k == -1LL) return -1;
        slot = (slot + 1u) & mask;
    }
    return -1;
}

__device__ __forceinline__ float aq3d_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int aq3d_wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// atomic max on a non-negative float through its bit pattern (monotone for v >= 0).
__device__ __forceinline__ void aq3d_atomic_max_pos(float *addr, float v) {
    atomicMax((unsigned int *)addr, __float_as_uint(v));
}
#endif
