// DIAGNOSTIC ONLY (scratch/diag). Second-level probe: which formulation of the rotated-overlap geometry stops being a pure
// function of its inputs next to other streams' kernels, and does it depend on the compiler's packed-fp32 lowering
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_mov_b32 with op_sel / neg modifiers)? This file is compiled TWICE into
// libaq3d_diag.so: as is (PROBE_TAG = pk) and with `-Xclang -target-feature -Xclang -packed-fp32-ops` (PROBE_TAG = nopk).
// counters[2*op] = in-launch mismatches (two evaluations of the same registers), counters[2*op+1] = vs the quiet launch.
#include "../../quantization-on-3d-object-detection_amd/csrc/nms_geom.h"
#include <stdint.h>

#ifndef PROBE_TAG
#define PROBE_TAG pk
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

namespace CAT(probe_, PROBE_TAG) {

__device__ __forceinline__ float seedf(unsigned h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return 0.5f + (float)(h >> 8) * (1.0f / 16777216.0f);
}

// branch-free restatement: everything computed, results selected
__device__ __forceinline__ bool seg_bf(const P2 &p1, const P2 &p0, const P2 &q1, const P2 &q0, P2 &ans) {
    const bool bb = bbox_overlap(p0, p1, q0, q1);
    const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    const bool ok = bb & (s1 * s2 > 0.f) & (s3 * s4 > 0.f);
    const float s5 = cross3(q1, p1, p0);
    const float den = s5 - s1;
    const float ax = (s5 * q0.x - s1 * q1.x) / den, ay = (s5 * q0.y - s1 * q1.y) / den;
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    const float bx = (b0 * c1 - b1 * c0) / D, by = (a1 * c0 - a0 * c1) / D;
    const bool ua = fabsf(den) > IOU_EPS;
    ans.x = ua ? ax : bx;
    ans.y = ua ? ay : by;
    return ok;
}

template <int OP>
__device__ float chain(float x, float y, float z, P2 *lds) {
    const float a[7] = {10.f * x, 5.f * y, 0.f, 0.3f + z, 40.f * x, 1.f, 2.9f * y};
    const float b[7] = {10.f * x + 0.4f * z, 5.f * y - 0.3f, 0.f, 0.2f + y, 55.f * z, 1.f, -1.3f * z};
    float acc = 0.f;
    if (OP == 0) return rotated_overlap<64>(a, b, lds);                        // control: the production function
    P2 ca[5], cb[5];
    box_corners(a, ca); box_corners(b, cb);
    if (OP == 1 || OP == 3) {                                                   // seg_intersection x16 (+ corner tests for 1)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { P2 ip = {0.f, 0.f}; if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], ip)) acc += ip.x * 0.5f + ip.y; }
    }
    if (OP == 1 || OP == 2) {                                                   // corner tests x8
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (corner_in_box(a, cb[k])) acc += cb[k].x; if (corner_in_box(b, ca[k])) acc -= ca[k].y; }
    }
    if (OP == 4) {                                                              // branch-free seg x16
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { P2 ip; const bool ok = seg_bf(ca[i + 1], ca[i], cb[j + 1], cb[j], ip); acc += ok ? ip.x * 0.5f + ip.y : 0.f; }
    }
    if (OP == 5) {                                                              // only the accept tests of seg_intersection (no divisions)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const P2 &p1 = ca[i + 1], &p0 = ca[i], &q1 = cb[j + 1], &q0 = cb[j];
                if (!bbox_overlap(p0, p1, q0, q1)) continue;
                const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
                if (s1 * s2 > 0.f && s3 * s4 > 0.f) acc += s1 + 0.5f * s3;
            }
    }
    if (OP == 6) {                                                              // only the divisions, unconditional
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const P2 &p1 = ca[i + 1], &p0 = ca[i], &q1 = cb[j + 1], &q0 = cb[j];
                const float s1 = cross3(q0, p1, p0), s5 = cross3(q1, p1, p0);
                acc += (s5 * q0.x - s1 * q1.x) / (s5 - s1 + 3.f) + (s5 * q0.y - s1 * q1.y) / (s5 - s1 + 3.f);
            }
    }
    if (OP == 7) {                                                              // one seg_intersection, 16 different lanes' worth of data
        P2 ip = {0.f, 0.f};
        if (seg_intersection(ca[1], ca[0], cb[2], cb[1], ip)) acc += ip.x * 0.5f + ip.y;
    }
    return acc;
}

template <int OP>
__device__ void probe(unsigned gid, P2 *lds, float *ref, int have_ref, int *counters) {
    float x = seedf(gid * 3u + 1u + 1000003u * OP), y = seedf(gid * 3u + 2u), z = seedf(gid * 3u + 3u) - 0.5f;
    const float r1 = chain<OP>(x, y, z, lds);
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z)::"memory");
    const float r2 = chain<OP>(x, y, z, lds);
    if (__float_as_int(r1) != __float_as_int(r2)) atomicAdd(&counters[2 * OP], 1);
    float *slot = ref + (size_t)OP * gridDim.x * 64 + gid;
    if (have_ref) { if (__float_as_int(*slot) != __float_as_int(r1)) atomicAdd(&counters[2 * OP + 1], 1); }
    else *slot = r1;
}

__global__ __launch_bounds__(64) void op_probe2_kernel(float *ref, int have_ref, int *counters) {
    __shared__ P2 lds_s[NMS_MAX_PTS * 64];
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    P2 *lds = lds_s + threadIdx.x;
    probe<0>(gid, lds, ref, have_ref, counters);
    probe<1>(gid, lds, ref, have_ref, counters);
    probe<2>(gid, lds, ref, have_ref, counters);
    probe<3>(gid, lds, ref, have_ref, counters);
    probe<4>(gid, lds, ref, have_ref, counters);
    probe<5>(gid, lds, ref, have_ref, counters);
    probe<6>(gid, lds, ref, have_ref, counters);
    probe<7>(gid, lds, ref, have_ref, counters);
    if (threadIdx.x == 0) { atomicAdd(&counters[30], 64); if (blockIdx.x == 0) atomicAdd(&counters[31], 1); }
}
// Discriminating variant: ONLY the two smallest formulations — few registers, no scratch in either build — so that the
// pk / nopk pair differs in instruction FORM alone, not in register footprint, occupancy or private-segment use.
__global__ __launch_bounds__(64) void op_probe2_small_kernel(float *ref, int have_ref, int *counters) {
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    probe<6>(gid, nullptr, ref, have_ref, counters);
    probe<7>(gid, nullptr, ref, have_ref, counters);
    if (threadIdx.x == 0) { atomicAdd(&counters[30], 64); if (blockIdx.x == 0) atomicAdd(&counters[31], 1); }
}
}  // namespace

extern "C" int CAT(CAT(aq3d_diag_op_probe2_, PROBE_TAG), _small)(float *ref, int32_t blocks, int32_t have_ref, int32_t *counters, void *stream) {
    hipLaunchKernelGGL(CAT(probe_, PROBE_TAG)::op_probe2_small_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, ref, have_ref, counters);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int CAT(aq3d_diag_op_probe2_, PROBE_TAG)(float *ref, int32_t blocks, int32_t have_ref, int32_t *counters, void *stream) {
    hipLaunchKernelGGL(CAT(probe_, PROBE_TAG)::op_probe2_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, ref, have_ref, counters);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
