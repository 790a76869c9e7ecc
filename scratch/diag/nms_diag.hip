// DIAGNOSTIC ONLY — built into scratch/diag/libaq3d_diag.so by scratch/diag/build_diag.py, never linked into or loaded by
// the product (libaq3d.so). Question it answers (VERDICT r1 "what's weak" 3): why do single bits of the NMS suppression mask
// differ for bit-identical candidate lists when other streams' kernels share the chip?
//
// One launch evaluates every (row, column block) word three ways and counts where the INPUTS the kernel saw differ:
//   E1  the production code path verbatim: row box through the compiler's uniform (scalar-cache, s_load) loads, column box
//       through plain vector loads (vector L1);
//   E2  the same function on boxes fetched with system-scope atomic loads (sc0 sc1: coherent, bypass K$ / L1);
//   E3  E2's inputs passed through opaque VGPR copies (no SGPR-resident operands, no scalar-unit arithmetic).
// counters[0] lane pairs evaluated            [1] lanes whose plain-loaded ROW box != coherent row box (bitwise)
//         [2] lanes whose plain-loaded COLUMN box != coherent column box
//         [3] mask words E1 != E2               [4] mask words E2 != E3
//         [5] lanes whose coherent row box changed between kernel start and kernel end (input rewritten under the kernel)
//         [6] same for the column box           [7] launches
#include "../../quantization-on-3d-object-detection_amd/csrc/nms_geom.h"
#include <stdint.h>

typedef unsigned long long u64;

__device__ __forceinline__ float coh_load(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void nms_diag_kernel(const float *__restrict__ boxes, const float *boxes_alias, int box_ld,
                                                      const int *count, int cap, float thresh, int col_blocks,
                                                      u64 *__restrict__ m1, u64 *__restrict__ m2, u64 *__restrict__ m3,
                                                      int *counters) {
    int n = *count;
    if (n > cap) n = cap;
    const int row = blockIdx.x, cb = blockIdx.y;
    if (row >= n) return;
    __shared__ P2 pts_s[NMS_MAX_PTS * 64];
    const int rb = row >> 6;
    u64 w1 = 0ull, w2 = 0ull, w3 = 0ull;
    if (cb >= rb) {
        const int j = cb * 64 + threadIdx.x;
        bool s1 = false, s2 = false, s3 = false;
        if (j < n && j > row) {
            float a[7], b[7], ac[7], bc[7], av[7], bv[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) { a[k] = boxes[(size_t)row * box_ld + k]; b[k] = boxes[(size_t)j * box_ld + k]; }
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                ac[k] = coh_load(boxes_alias + (size_t)row * box_ld + k);
                bc[k] = coh_load(boxes_alias + (size_t)j * box_ld + k);
                av[k] = ac[k]; bv[k] = bc[k];
                asm volatile("" : "+v"(av[k]), "+v"(bv[k]));
            }
            bool da = false, db = false;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                da = da || __float_as_int(a[k]) != __float_as_int(ac[k]);
                db = db || __float_as_int(b[k]) != __float_as_int(bc[k]);
            }
            atomicAdd(&counters[0], 1);
            if (da) atomicAdd(&counters[1], 1);
            if (db) atomicAdd(&counters[2], 1);
            if (circles_may_touch(a, b)) s1 = iou_bev_dev<64>(a, b, pts_s + threadIdx.x) > thresh;
            asm volatile("" ::: "memory");
            if (circles_may_touch(ac, bc)) s2 = iou_bev_dev<64>(ac, bc, pts_s + threadIdx.x) > thresh;
            asm volatile("" ::: "memory");
            if (circles_may_touch(av, bv)) s3 = iou_bev_dev<64>(av, bv, pts_s + threadIdx.x) > thresh;
            bool ca = false, cbx = false;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                ca = ca || __float_as_int(coh_load(boxes_alias + (size_t)row * box_ld + k)) != __float_as_int(ac[k]);
                cbx = cbx || __float_as_int(coh_load(boxes_alias + (size_t)j * box_ld + k)) != __float_as_int(bc[k]);
            }
            if (ca) atomicAdd(&counters[5], 1);
            if (cbx) atomicAdd(&counters[6], 1);
        }
        w1 = __ballot(s1); w2 = __ballot(s2); w3 = __ballot(s3);
    }
    if (threadIdx.x == 0) {
        m1[(size_t)row * col_blocks + cb] = w1;
        m2[(size_t)row * col_blocks + cb] = w2;
        m3[(size_t)row * col_blocks + cb] = w3;
        if (w1 != w2) atomicAdd(&counters[3], 1);
        if (w2 != w3) atomicAdd(&counters[4], 1);
        if (row == 0 && cb == 0) atomicAdd(&counters[7], 1);
    }
}

extern "C" int aq3d_diag_nms_masks(const float *boxes, int32_t box_ld, const int32_t *count, int32_t cap, float thresh,
                                   u64 *m1, u64 *m2, u64 *m3, int32_t *counters, void *stream) {
    const int col_blocks = (cap + 63) / 64;
    hipLaunchKernelGGL(nms_diag_kernel, dim3(cap, col_blocks), dim3(64), 0, (hipStream_t)stream, boxes, boxes, box_ld, count, cap,
                       thresh, col_blocks, m1, m2, m3, counters);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
