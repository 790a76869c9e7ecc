// CenterHead decode (top-K over the heat map, box assembly, range/score mask) and rotated BEV NMS, all on device.
//
// Reference semantics:
//   decode  — pcdet/models/model_utils/centernet_utils.py:155-241 (_topk: per-class top-K over H*W, then top-K over
//             classes*K; gather; atan2; x = (xs + dx) * stride * voxel + range_min; range + score mask),
//             called from pcdet/models/dense_heads/center_head.py:297-330 (sigmoid on hm, exp on dim).
//   NMS     — pcdet/models/model_utils/model_nms_utils.py:6-25 → pcdet/ops/iou3d_nms/iou3d_nms_utils.py:120-135 →
//             pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:137-183 (bitmask + greedy sweep) with the rotated-rectangle IoU of
//             pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:35-234 (== iou3d_cpu.cpp:59-229).
// Differences by design: the reference copies the N x N/64 bitmask to the host and sweeps it on the CPU
// (one blocking D2H per head per frame); here one wave sweeps it on device, 64 boxes per step. torch.topk's order
// among equal scores is unspecified; we fix it to (score desc, flat index asc) and the oracle does the same.
#include "common.h"
#include "nms_geom.h"
#include <stdlib.h>
#include <stdio.h>

typedef unsigned long long u64;

// ------------------------------------------------------------------ block-wide helpers (1024 threads)
__device__ __forceinline__ int block_scan_excl_1024(int v, int *sh, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
        int s = sh[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// in-LDS bitonic sort, descending, n = power of two, any block size
__device__ __forceinline__ void bitonic_sort_desc(u64 *keys, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    u64 a = keys[i], b = keys[ixj];
                    bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) {
                        keys[i] = b;
                        keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Same sort, far fewer barriers: every wave keeps 128 consecutive keys in registers (lane l: elements base + l and
// base + 64 + l); compare-exchange distances below 64 are wave shuffles, distance 64 is a swap inside the lane, and only
// distances >= 128 go through LDS behind a barrier. For n = 2048 that is 10 barrier rounds instead of 66 — the decode
// stage is a single-workgroup latency chain at the end of the frame.
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    const unsigned lo = __shfl_xor((unsigned)v, m, 64), hi = __shfl_xor((unsigned)(v >> 32), m, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ void bitonic_sort_desc_hybrid(u64 *keys, int n) {       // n: power of two, 128 <= n <= 128 * waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = wave * 128;
    const bool active = base < n;
    u64 e0 = active ? keys[base + lane] : 0ull, e1 = active ? keys[base + 64 + lane] : 0ull;
    auto local_stages = [&](int k, int jmax) {       // sub-stages j = jmax .. 1 (jmax <= 64) of merge size k, in registers
        for (int j = jmax; j > 0; j >>= 1) {
            if (j == 64) {
                const bool desc = ((base + lane) & k) == 0;
                const bool sw = desc ? (e0 < e1) : (e0 > e1);
                const u64 a = sw ? e1 : e0, b = sw ? e0 : e1;
                e0 = a; e1 = b;
            } else {
                const u64 p0 = shfl_xor_u64(e0, j), p1 = shfl_xor_u64(e1, j);
                const bool upper = (lane & j) != 0;           // this lane holds the element with the larger index of the pair
                const bool d0 = ((base + lane) & k) == 0, d1 = ((base + 64 + lane) & k) == 0;
                // descending block: the lower index keeps the larger key
                e0 = (d0 != upper) ? (e0 > p0 ? e0 : p0) : (e0 < p0 ? e0 : p0);
                e1 = (d1 != upper) ? (e1 > p1 ? e1 : p1) : (e1 < p1 ? e1 : p1);
            }
        }
    };
    for (int k = 2; k <= n; k <<= 1) {
        int j = k >> 1;
        if (j >= 128) {
            // distances >= 128: through LDS, one barrier pair per sub-stage
            if (active) { keys[base + lane] = e0; keys[base + 64 + lane] = e1; }
            __syncthreads();
            for (; j >= 128; j >>= 1) {
                for (int i = threadIdx.x; i < n; i += blockDim.x) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const u64 a = keys[i], b = keys[ixj];
                        const bool desc = (i & k) == 0;
                        if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[ixj] = a; }
                    }
                }
                __syncthreads();
            }
            if (active) { e0 = keys[base + lane]; e1 = keys[base + 64 + lane]; }
        }
        if (active) local_stages(k, j < 64 ? j : 64);
    }
    if (active) { keys[base + lane] = e0; keys[base + 64 + lane] = e1; }
    __syncthreads();
}

__device__ __forceinline__ float sigmoidf_dev(float x) { return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x))); }

// ------------------------------------------------------------------ stage 1: per-class top-K (radix select + sort)
// grid = num_classes, block = 1024. Writes K composite keys (score bits << 32 | ~index) per class, sorted descending;
// slots beyond the number of pixels hold 0.
__global__ __launch_bounds__(1024) void topk_per_class_kernel(const float *__restrict__ hm, int hm_ld, int L_static,
                                                              const int *__restrict__ count, int K, int Kpad,
                                                              int cache_keys, float score_thresh, u64 *__restrict__ out) {
    int L = L_static;            // dense map: H*W pixels; sparse head: the live rows (device count, capped by L_static)
    if (count) { L = *count; if (L > L_static) L = L_static; if (L < 0) L = 0; }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *sel = reinterpret_cast<u64 *>(smem_raw);          // [Kpad]
    int *hist = reinterpret_cast<int *>(sel + Kpad);       // [256]
    int *sh = hist + 256;                                  // [16]
    unsigned int *keys_s = reinterpret_cast<unsigned int *>(sh + 16);   // [L] score bit patterns (when they fit in LDS)
    constexpr int TIE_CAP = 512;
    __shared__ unsigned int s_prefix;
    __shared__ int s_remaining, s_fill, s_ties;
    __shared__ unsigned int tie_idx[TIE_CAP];
    const int cls = blockIdx.x;
    const float *col = hm + cls;
    auto key_of = [&](int i) -> unsigned int {
        return cache_keys ? keys_s[i] : __float_as_uint(sigmoidf_dev(col[(size_t)i * hm_ld]));
    };
    // Pre-filter: a candidate whose score is not above score_thresh is masked by the decode stage whatever its rank, so
    // the top-K only has to be found among the scores above it (if fewer than K exist, the missing ones would be masked
    // anyway). An ORDER-PRESERVING compaction (each wave owns a contiguous pixel range; counts, a 16-wave prefix, then
    // ballot-ranked writes) shrinks the radix-select domain from H*W to the few thousand live peaks and keeps every tie
    // rule (smaller pixel index first) intact. Falls back to the full map when more than FCAP scores pass.
    constexpr int FPER = 36, FCAP = 16384;
    unsigned int *idx_s = keys_s + FCAP;                   // filtered mode: [FCAP] original pixel indices
    __shared__ int s_wcnt[16], s_filtered;
    const bool try_filter = cache_keys && score_thresh >= 0.f && L <= FPER * 1024 && blockDim.x == 1024;
    if (threadIdx.x == 0) s_filtered = 0;
    if (try_filter) {
        const unsigned int tkey = __float_as_uint(score_thresh);
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int chunk = (((L + 15) >> 4) + 63) & ~63;     // pixels per wave, multiple of 64
        const int beg = w * chunk, end = min(L, beg + chunk);
        unsigned int kv[FPER];
        int cnt = 0;
#pragma unroll
        for (int it = 0; it < FPER; ++it) {
            const int i = beg + it * 64 + lane;
            kv[it] = (it * 64 < chunk && i < end) ? __float_as_uint(sigmoidf_dev(col[(size_t)i * hm_ld])) : 0u;
            cnt += (int)__popcll(__ballot(kv[it] > tkey));
        }
        if (lane == 0) s_wcnt[w] = cnt;
        __syncthreads();
        int base = 0, total = 0;
        for (int q = 0; q < 16; ++q) { base += q < w ? s_wcnt[q] : 0; total += s_wcnt[q]; }
        if (total <= FCAP) {
#pragma unroll
            for (int it = 0; it < FPER; ++it) {
                const bool pass_ = kv[it] > tkey;
                const u64 m = __ballot(pass_);
                if (pass_) {
                    const int pos = base + (int)__popcll(m & ((1ull << lane) - 1ull));
                    keys_s[pos] = kv[it];
                    idx_s[pos] = (unsigned int)(beg + it * 64 + lane);
                }
                base += (int)__popcll(m);
            }
            if (threadIdx.x == 0) s_filtered = 1;
            L = total;
        }
        __syncthreads();
    }
    const bool filtered = try_filter && s_filtered;
    const int Keff = K < L ? K : L;
    auto orig = [&](int i) -> unsigned int { return filtered ? idx_s[i] : (unsigned int)i; };
    if (cache_keys && !filtered) {
        // strided column read (one float per pixel row): 8 independent loads in flight per thread
        const int nth = blockDim.x;
        for (int base = threadIdx.x; base < L; base += 8 * nth) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * nth;
                v[u] = i < L ? col[(size_t)i * hm_ld] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * nth;
                if (i < L) keys_s[i] = __float_as_uint(sigmoidf_dev(v[u]));
            }
        }
    }

    // 4-pass MSB radix select on the 32-bit score keys: find T = Keff-th largest key
    if (threadIdx.x == 0) { s_prefix = 0u; s_remaining = Keff; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        const unsigned int prefix = s_prefix;
        const unsigned int pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        for (int base = 0; base < L; base += blockDim.x) {
            const int i = base + threadIdx.x;
            unsigned int bin = 0xffffffffu;
            if (i < L) {
                const unsigned int key = key_of(i);
                if ((key & pmask) == prefix) bin = (key >> shift) & 0xff;
            }
            if (pass > 0) {   // lower key bits are spread out: plain LDS atomics
                if (bin != 0xffffffffu) atomicAdd(&hist[bin], 1);
                continue;
            }
            // top byte of a heat-map score: nearly every key shares it → one atomic per wave and distinct value
            u64 active = __ballot(bin != 0xffffffffu);
            while (active) {
                const int leader = __ffsll((long long)active) - 1;
                const unsigned int lb = __shfl(bin, leader, 64);
                const u64 same = __ballot(bin == lb);
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lb], (int)__popcll(same));
                active &= ~same;
            }
        }
        __syncthreads();
        // suffix sums of the 256 bins by the first 256 threads (4 waves): bin b is the pivot iff
        // sum(hist[b+1..255]) < rem <= sum(hist[b..255])
        {
            int v = 0, incl = 0;
            if (threadIdx.x < 256) {
                v = hist[255 - threadIdx.x];              // reversed: prefix sums from the top bin down
                incl = v;
                const int lane = threadIdx.x & 63;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    int t = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += t;
                }
                if (lane == 63) sh[threadIdx.x >> 6] = incl;
            }
            // read BEFORE the barrier: the pivot thread below overwrites s_remaining, and a wave that arrives later must not
            // test against the new value (it could elect a second pivot — seen as one detection more or less, once in a few
            // hundred frames, when another stream's kernels shared the CU and pulled the four waves apart)
            const int rem = s_remaining;
            __syncthreads();
            if (threadIdx.x < 256) {
                int basev = 0;
                for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) basev += sh[w];
                incl += basev;                             // keys in bins >= (255 - tid)
                if (incl >= rem && incl - v < rem) {       // exactly one thread
                    s_prefix = prefix | ((unsigned int)(255 - threadIdx.x) << shift);
                    s_remaining = rem - (incl - v);
                }
            }
            __syncthreads();
        }
    }
    const unsigned int T = s_prefix;
    const int need_ties = s_remaining;
    for (int i = threadIdx.x; i < Kpad; i += blockDim.x) sel[i] = 0ull;
    if (threadIdx.x == 0) { s_fill = 0; s_ties = 0; }
    __syncthreads();
    // keys > T in any order (sorted afterwards); keys == T: collect up to TIE_CAP indices
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const unsigned int key = key_of(i);
        if (key > T) {
            int pos = atomicAdd(&s_fill, 1);
            sel[pos] = ((u64)key << 32) | (u64)(0xffffffffu - orig(i));
        } else if (key == T) {
            int pos = atomicAdd(&s_ties, 1);
            if (pos < TIE_CAP) tie_idx[pos] = orig(i);
        }
    }
    __syncthreads();
    const int gt = s_fill, nties = s_ties;
    if (nties <= TIE_CAP) {
        // few ties (the normal case): rank each tie by its index, the need_ties smallest go in
        for (int t = threadIdx.x; t < nties; t += blockDim.x) {
            const unsigned int mine_i = tie_idx[t];
            int rank = 0;
            for (int q = 0; q < nties; ++q) rank += tie_idx[q] < mine_i;
            if (rank < need_ties) sel[gt + rank] = ((u64)T << 32) | (u64)(0xffffffffu - mine_i);
        }
    } else {
        // many equal scores (saturated / constant heat map): ordered compaction, 1024 pixels per round
        int taken = 0;
        for (int base = 0; base < L && taken < need_ties; base += blockDim.x) {
            const int i = base + threadIdx.x;
            int f = 0;
            if (i < L) f = (key_of(i) == T) ? 1 : 0;
            int tot;
            int pos = block_scan_excl_1024(f, sh, &tot);
            if (f && taken + pos < need_ties) sel[gt + taken + pos] = ((u64)T << 32) | (u64)(0xffffffffu - orig(i));
            taken += tot;
        }
    }
    __syncthreads();
    // unsorted: stage 2 orders everything by (score desc, class asc, pixel asc) itself
    for (int i = threadIdx.x; i < K; i += blockDim.x) out[(size_t)cls * K + i] = (i < Keff) ? sel[i] : 0ull;
}

// ------------------------------------------------------------------ stage 2: top-K over classes*K, decode, mask
struct DecodeParams {
    aq3d_decode_cfg c;
    const float *center, *center_z, *dim, *rot, *vel;
    const int *coords;      // sparse head: rows (b, z|0, y, x) of the voxels; NULL for a dense map
};

__global__ __launch_bounds__(1024) void decode_kernel(const u64 *__restrict__ cand, int ncand, int Npad, DecodeParams q,
                                                      float *__restrict__ boxes_out, float *__restrict__ scores_out,
                                                      int *__restrict__ labels_out, int *__restrict__ count_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *keys = reinterpret_cast<u64 *>(smem_raw);  // [Npad] : score bits << 32 | ~position (position = class*K + rank)
    __shared__ int sh[16];
    const int K = q.c.K;
    for (int i = threadIdx.x; i < Npad; i += blockDim.x) {
        u64 v = 0ull;
        if (i < ncand) {
            const u64 cv = cand[i];
            // ties: class-major, then pixel index ascending — what torch.topk over the [ncls, K] matrix of per-class
            // lists (each ordered by score desc / pixel asc) yields; independent of the slot order inside a class
            const unsigned pixel = 0xffffffffu - (unsigned int)(cv & 0xffffffffull);
            v = (cv & 0xffffffff00000000ull) | (u64)(0xffffffffu - (((unsigned int)(i / K) << 27) | pixel));
            if (cv == 0ull) v = 0ull;  // empty slot
        }
        keys[i] = v;
    }
    __syncthreads();
    if (Npad >= 128 && Npad <= 2048 && blockDim.x == 1024) bitonic_sort_desc_hybrid(keys, Npad);
    else bitonic_sort_desc(keys, Npad);

    const int nbox = q.c.has_vel ? 9 : 7;
    const int t = threadIdx.x;
    int valid = 0;
    float box[9];
    float score = 0.f;
    int label = 0;
    if (t < K && keys[t] != 0ull) {
        const u64 kv = keys[t];
        const unsigned cp = 0xffffffffu - (unsigned int)(kv & 0xffffffffull);
        label = (int)(cp >> 27);
        const int pix = (int)(cp & 0x07ffffffu);
        score = __uint_as_float((unsigned int)(kv >> 32));
        int yi, xi;
        if (q.coords) { yi = q.coords[(size_t)pix * 4 + 2]; xi = q.coords[(size_t)pix * 4 + 3]; }
        else { yi = pix / q.c.W; xi = pix - yi * q.c.W; }
        const size_t ro = (size_t)pix * q.c.reg_ld;
        const float cx = q.center[ro + 0], cy = q.center[ro + 1];
        float xs = __fadd_rn((float)xi, cx), ys = __fadd_rn((float)yi, cy);
        xs = __fadd_rn(__fmul_rn(__fmul_rn(xs, q.c.feature_map_stride), q.c.voxel_size[0]), q.c.pc_range_min[0]);
        ys = __fadd_rn(__fmul_rn(__fmul_rn(ys, q.c.feature_map_stride), q.c.voxel_size[1]), q.c.pc_range_min[1]);
        box[0] = xs;
        box[1] = ys;
        box[2] = q.center_z[ro];
        box[3] = expf(q.dim[ro + 0]);
        box[4] = expf(q.dim[ro + 1]);
        box[5] = expf(q.dim[ro + 2]);
        box[6] = atan2f(q.rot[ro + 1], q.rot[ro + 0]);  // rot = (cos, sin)
        box[7] = box[8] = 0.f;
        if (q.c.has_vel) { box[7] = q.vel[ro + 0]; box[8] = q.vel[ro + 1]; }
        valid = 1;
#pragma unroll
        for (int j = 0; j < 3; ++j) valid &= (box[j] >= q.c.post_center_range[j]) && (box[j] <= q.c.post_center_range[3 + j]);
        if (q.c.score_thresh >= 0.f) valid &= score > q.c.score_thresh;
    }
    int total;
    const int dst = block_scan_excl_1024(valid, sh, &total);
    if (valid) {
        for (int j = 0; j < nbox; ++j) boxes_out[(size_t)dst * nbox + j] = box[j];
        scores_out[dst] = score;
        labels_out[dst] = label;
    }
    if (threadIdx.x == 0) *count_out = total;
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

extern "C" size_t aq3d_center_decode_workspace_bytes(const aq3d_decode_cfg *cfg) {
    if (!cfg) return 0;
    return (size_t)cfg->num_classes * cfg->K * sizeof(u64) + 256;
}

extern "C" int aq3d_center_decode(const float *hm, const float *center, const float *center_z, const float *dim,
                                  const float *rot, const float *vel, const aq3d_decode_cfg *cfg, float *boxes_out,
                                  float *scores_out, int32_t *labels_out, int32_t *count_out, void *workspace,
                                  size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(hm && center && center_z && dim && rot && cfg && boxes_out && scores_out && labels_out && count_out && workspace,
                 "aq3d_center_decode: null argument");
    AQ3D_REQUIRE(!cfg->has_vel || vel, "aq3d_center_decode: has_vel without vel map");
    AQ3D_REQUIRE(cfg->num_classes >= 1 && cfg->num_classes <= 16, "aq3d_center_decode: num_classes=%d", cfg->num_classes);
    AQ3D_REQUIRE(cfg->K >= 1 && cfg->K <= 512, "aq3d_center_decode: K=%d not in 1..512", cfg->K);
    AQ3D_REQUIRE(cfg->H > 0 && cfg->W > 0, "aq3d_center_decode: empty map");
    if (workspace_bytes < aq3d_center_decode_workspace_bytes(cfg)) {
        aq3d_set_error("aq3d_center_decode: workspace too small");
        return AQ3D_ERR_WORKSPACE;
    }
    const int L = cfg->H * cfg->W, K = cfg->K;
    const int Kpad = next_pow2(K);
    u64 *cand = (u64 *)workspace;
    size_t lds1 = (size_t)Kpad * 8 + 256 * 4 + 16 * 4;
    const bool no_filter = false;     // the order-preserving score-threshold pre-filter is always on (identical detections)
    // key cache [L] — or, in the pre-filtered mode, keys + pixel indices of up to 16384 candidates (2 x 64 KiB)
    const size_t key_bytes = (size_t)L * 4 > 131072 ? (size_t)L * 4 : 131072;
    const int cache_keys = (lds1 + key_bytes <= 150 * 1024) ? 1 : 0;   // one workgroup may use up to 160 KiB of LDS
    if (cache_keys) lds1 += key_bytes;
    if (lds1 > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(topk_per_class_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    hipLaunchKernelGGL(topk_per_class_kernel, dim3(cfg->num_classes), dim3(1024), lds1, stream, hm, cfg->hm_ld, L, (const int *)nullptr,
                       K, Kpad, cache_keys, no_filter ? -1.f : cfg->score_thresh, cand);
    const int ncand = cfg->num_classes * K;
    const int Npad = next_pow2(ncand);
    DecodeParams q;
    q.c = *cfg; q.center = center; q.center_z = center_z; q.dim = dim; q.rot = rot; q.vel = vel; q.coords = nullptr;
    size_t lds2 = (size_t)Npad * 8;
    if (lds2 > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipLaunchKernelGGL(decode_kernel, dim3(1), dim3(1024), lds2, stream, cand, ncand, Npad, q, boxes_out, scores_out, labels_out, count_out);
    AQ3D_CHECK_LAUNCH("aq3d_center_decode");
    return AQ3D_OK;
}

extern "C" int aq3d_voxel_decode(const float *hm, const float *center, const float *center_z, const float *dim,
                                 const float *rot, const float *vel, const int32_t *coords, const int32_t *count, int32_t cap,
                                 const aq3d_decode_cfg *cfg, float *boxes_out, float *scores_out, int32_t *labels_out,
                                 int32_t *count_out, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(hm && center && center_z && dim && rot && coords && count && cfg && boxes_out && scores_out && labels_out &&
                     count_out && workspace, "aq3d_voxel_decode: null argument");
    AQ3D_REQUIRE(!cfg->has_vel || vel, "aq3d_voxel_decode: has_vel without vel rows");
    AQ3D_REQUIRE(cfg->num_classes >= 1 && cfg->num_classes <= 16 && cfg->K >= 1 && cfg->K <= 512 && cap >= 1,
                 "aq3d_voxel_decode: num_classes=%d K=%d cap=%d", cfg->num_classes, cfg->K, cap);
    if (workspace_bytes < aq3d_center_decode_workspace_bytes(cfg)) {
        aq3d_set_error("aq3d_voxel_decode: workspace too small");
        return AQ3D_ERR_WORKSPACE;
    }
    const int K = cfg->K, Kpad = next_pow2(K);
    u64 *cand = (u64 *)workspace;
    size_t lds1 = (size_t)Kpad * 8 + 256 * 4 + 16 * 4;
    const size_t key_bytes = (size_t)cap * 4 > 131072 ? (size_t)cap * 4 : 131072;
    const int cache_keys = (lds1 + key_bytes <= 150 * 1024) ? 1 : 0;
    if (cache_keys) lds1 += key_bytes;
    if (lds1 > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(topk_per_class_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    hipLaunchKernelGGL(topk_per_class_kernel, dim3(cfg->num_classes), dim3(1024), lds1, stream, hm, cfg->hm_ld, cap, count, K, Kpad,
                       cache_keys, cfg->score_thresh, cand);
    const int ncand = cfg->num_classes * K, Npad = next_pow2(ncand);
    DecodeParams q;
    q.c = *cfg; q.center = center; q.center_z = center_z; q.dim = dim; q.rot = rot; q.vel = vel; q.coords = coords;
    size_t lds2 = (size_t)Npad * 8;
    if (lds2 > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipLaunchKernelGGL(decode_kernel, dim3(1), dim3(1024), lds2, stream, cand, ncand, Npad, q, boxes_out, scores_out, labels_out, count_out);
    AQ3D_CHECK_LAUNCH("aq3d_voxel_decode");
    return AQ3D_OK;
}

// ------------------------------------------------------------------ rotated NMS (geometry: nms_geom.h)
// one wave per (row box, column block): lane j tests box (cb*64 + j) against the row box, ballot -> one mask word
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, int box_ld, const int *count, int cap,
                                                      float thresh, int col_blocks, u64 *__restrict__ mask) {
    int n = *count;
    if (n > cap) n = cap;
    const int row = blockIdx.x, cb = blockIdx.y;
    if (row >= n) return;
    __shared__ P2 pts_s[NMS_MAX_PTS * 64];
    const int rb = row >> 6;
    u64 word = 0ull;
    if (cb >= rb) {
        const int j = cb * 64 + threadIdx.x;
        bool sup = false;
        if (j < n && j > row) {
            float a[7], b[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) { a[k] = boxes[(size_t)row * box_ld + k]; b[k] = boxes[(size_t)j * box_ld + k]; }
            if (circles_may_touch(a, b)) sup = iou_bev_dev<64>(a, b, pts_s + threadIdx.x) > thresh;
        }
        word = __ballot(sup);
    }
    if (threadIdx.x == 0) mask[(size_t)row * col_blocks + cb] = word;
}


// greedy sweep, 64 boxes per step; the bitmask is staged in LDS by the whole workgroup when it fits, then wave 0 sweeps:
// lane w owns removal word w (cap <= 4096 → <= 64 words).
__global__ __launch_bounds__(256) void nms_sweep_kernel(const u64 *__restrict__ mask, const int *count, int cap, int col_blocks,
                                                       int post_max, int use_lds, int *__restrict__ keep, int *__restrict__ keep_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *mask_s = reinterpret_cast<u64 *>(smem_raw);
    int n = *count;
    if (n > cap) n = cap;
    const u64 *msrc = mask;
    if (use_lds) {
        const int words = n * col_blocks;
        for (int i = threadIdx.x; i < words; i += blockDim.x) mask_s[i] = mask[i];
        __syncthreads();
        msrc = mask_s;
    }
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    u64 remv = 0ull;
    int num = 0;
    const int nb = (n + 63) >> 6;
    for (int b = 0; b < nb; ++b) {
        const int row = b * 64 + lane;
        const u64 diag = (row < n) ? msrc[(size_t)row * col_blocks + b] : 0ull;
        // the greedy chain is wave-uniform: keep it on the scalar unit (v_readlane + 64-bit SALU ops) instead of 64-wide
        // shuffles and VALU bit ops that every lane repeats
        auto bcast = [&](u64 v, int src) -> u64 {
            const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, src);
            const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
            return ((u64)hi << 32) | lo;
        };
        u64 cur = bcast(remv, b);
        u64 kept = 0ull;
        const int lim = min(64, n - b * 64);
        for (int t = 0; t < lim; ++t) {
            const u64 d = bcast(diag, t);
            const bool take = !((cur >> t) & 1ull);
            if (take) { kept |= 1ull << t; cur |= d; }
        }
        if ((kept >> lane) & 1ull) {
            const int pos = num + __popcll(kept & ((1ull << lane) - 1ull));
            if (pos < post_max) keep[pos] = row;
        }
        num += __popcll(kept);
        if (lane > b && lane < col_blocks) {
#pragma unroll 8
            for (int t = 0; t < 64; ++t) {
                const int r = b * 64 + t;
                const u64 m = (r < n) ? msrc[(size_t)r * col_blocks + lane] : 0ull;
                if ((kept >> t) & 1ull) remv |= m;
            }
        }
    }
    if (lane == 0) *keep_count = num < post_max ? num : post_max;
}

extern "C" size_t aq3d_nms_workspace_bytes(int32_t cap) {
    if (cap < 1) cap = 1;
    return (size_t)cap * aq3d_div_up(cap, 64) * sizeof(u64) + 256;
}

extern "C" int aq3d_nms_rotated(const float *boxes, int32_t box_ld, const int32_t *count, int32_t cap, float thresh,
                                int32_t post_max, int32_t *keep_out, int32_t *keep_count, void *workspace,
                                size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(boxes && count && keep_out && keep_count && workspace, "aq3d_nms_rotated: null argument");
    AQ3D_REQUIRE(cap >= 1 && cap <= 4096 && box_ld >= 7 && post_max >= 1, "aq3d_nms_rotated: cap=%d box_ld=%d post_max=%d", cap, box_ld, post_max);
    if (workspace_bytes < aq3d_nms_workspace_bytes(cap)) {
        aq3d_set_error("aq3d_nms_rotated: workspace too small");
        return AQ3D_ERR_WORKSPACE;
    }
    const int col_blocks = aq3d_div_up(cap, 64);
    u64 *mask = (u64 *)workspace;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cap, col_blocks), dim3(64), 0, stream, boxes, box_ld, count, cap, thresh, col_blocks, mask);
    const size_t mask_bytes = (size_t)cap * col_blocks * sizeof(u64);
    const int use_lds = mask_bytes <= 128 * 1024;
    if (use_lds && mask_bytes > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(nms_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mask_bytes);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(256), use_lds ? mask_bytes : 0, stream, mask, count, cap, col_blocks, post_max,
                       use_lds, keep_out, keep_count);
    AQ3D_CHECK_LAUNCH("aq3d_nms_rotated");
    return AQ3D_OK;
}

__global__ __launch_bounds__(256) void gather_dets_kernel(const float *__restrict__ boxes, int box_ld,
                                                          const float *__restrict__ scores, const int *__restrict__ labels,
                                                          const int *__restrict__ class_map, const int *__restrict__ keep,
                                                          const int *keep_count, int cap, float *__restrict__ boxes_out,
                                                          float *__restrict__ scores_out, long long *__restrict__ labels_out) {
    int n = *keep_count;
    if (n > cap) n = cap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int src = keep[i];
        for (int j = 0; j < box_ld; ++j) boxes_out[(size_t)i * box_ld + j] = boxes[(size_t)src * box_ld + j];
        scores_out[i] = scores[src];
        const int lab = labels[src];
        labels_out[i] = (long long)(class_map ? class_map[lab] : lab) + 1;  // center_head.py:363
    }
}

extern "C" int aq3d_gather_dets(const float *boxes, int32_t box_ld, const float *scores, const int32_t *labels,
                                const int32_t *class_map, const int32_t *keep, const int32_t *keep_count, int32_t cap,
                                float *boxes_out, float *scores_out, int64_t *labels_out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(boxes && scores && labels && keep && keep_count && boxes_out && scores_out && labels_out,
                 "aq3d_gather_dets: null argument");
    hipLaunchKernelGGL(gather_dets_kernel, dim3(aq3d_grid(cap, 256)), dim3(256), 0, stream, boxes, box_ld, scores, labels,
                       class_map, keep, keep_count, cap, boxes_out, scores_out, (long long *)labels_out);
    AQ3D_CHECK_LAUNCH("aq3d_gather_dets");
    return AQ3D_OK;
}

// ------------------------------------------------------------------ 3-D IoU matrix + recall counts
__global__ __launch_bounds__(256) void boxes_iou3d_kernel(const float *__restrict__ A, int n, int lda,
                                                          const float *__restrict__ B, int m, int ldb,
                                                          float *__restrict__ out) {
    const long long total = (long long)n * m;
    __shared__ P2 pts_s[NMS_MAX_PTS * 256];
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int i = (int)(t / m), j = (int)(t - (long long)i * m);
        float a[7], b[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) { a[k] = A[(size_t)i * lda + k]; b[k] = B[(size_t)j * ldb + k]; }
        // iou3d_nms_utils.py:58-79 — torch ops there, same order here
        const float a_max = a[2] + a[5] / 2, a_min = a[2] - a[5] / 2, b_max = b[2] + b[5] / 2, b_min = b[2] - b[5] / 2;
        const float ov_bev = rotated_overlap<256>(a, b, pts_s + threadIdx.x);
        const float ov_h = fmaxf(fminf(a_max, b_max) - fmaxf(a_min, b_min), 0.f);
        const float ov3d = ov_bev * ov_h;
        const float va = a[3] * a[4] * a[5], vb = b[3] * b[4] * b[5];
        out[t] = ov3d / fmaxf(va + vb - ov3d, 1e-6f);
    }
}

extern "C" int aq3d_boxes_iou3d(const float *boxes_a, int32_t n, int32_t lda, const float *boxes_b, int32_t m, int32_t ldb,
                                float *iou_out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(n >= 0 && m >= 0 && lda >= 7 && ldb >= 7, "aq3d_boxes_iou3d: bad sizes n=%d m=%d lda=%d ldb=%d", n, m, lda, ldb);
    if (n == 0 || m == 0) return AQ3D_OK;
    AQ3D_REQUIRE(boxes_a && boxes_b && iou_out, "aq3d_boxes_iou3d: null argument");
    hipLaunchKernelGGL(boxes_iou3d_kernel, dim3(aq3d_grid((int64_t)n * m, 256)), dim3(256), 0, stream, boxes_a, n, lda, boxes_b, m,
                       ldb, iou_out);
    AQ3D_CHECK_LAUNCH("aq3d_boxes_iou3d");
    return AQ3D_OK;
}

__global__ __launch_bounds__(256) void recall_counts_kernel(const float *__restrict__ iou, int n, int m,
                                                            const float *__restrict__ thr, int nt, int *counts) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256) {
        float mx = -1.f;
        for (int i = 0; i < n; ++i) mx = fmaxf(mx, iou[(size_t)i * m + j]);
        for (int t = 0; t < nt; ++t)
            if (mx > thr[t]) atomicAdd(&counts[t], 1);
    }
}

extern "C" int aq3d_recall_counts(const float *iou, int32_t n, int32_t m, const float *thresholds, int32_t n_thresh,
                                  int32_t *counts_out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(n >= 0 && m >= 0 && n_thresh >= 1, "aq3d_recall_counts: bad sizes");
    if (n == 0 || m == 0) return AQ3D_OK;
    AQ3D_REQUIRE(iou && thresholds && counts_out, "aq3d_recall_counts: null argument");
    hipLaunchKernelGGL(recall_counts_kernel, dim3(aq3d_grid(m, 256)), dim3(256), 0, stream, iou, n, m, thresholds, n_thresh,
                       counts_out);
    AQ3D_CHECK_LAUNCH("aq3d_recall_counts");
    return AQ3D_OK;
}
