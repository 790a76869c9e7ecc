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

// In-kernel clock stamps of the single-workgroup kernels of this file for scratch/dn_stamps.py (diagnostic build with -DAQ3D_STAMPS
// only; the product library compiles none of it): thread 0 of the first workgroup writes clock64() into slot [kernel][i].
#ifdef AQ3D_STAMPS
__device__ unsigned long long *g_dn_stamps;          // [4 kernels][16]
extern "C" int aq3d_diag_set_dn_stamps(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dn_stamps), &p, sizeof(p)); }
#define DN_STAMP(kernel, i) do { if (g_dn_stamps && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_dn_stamps[(kernel) * 16 + (i)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define DN_STAMP(kernel, i)
#endif

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
__device__ __forceinline__ void bitonic_sort_desc_hybrid(u64 *keys, int n) {       // n: power of two, 128 <= n <= 128 * waves (<= 2048)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = wave * 128;
    const bool active = base < n;
    u64 e0 = active ? keys[base + lane] : 0ull, e1 = active ? keys[base + 64 + lane] : 0ull;
    // compare-exchange of a lane's key with its partner's: the lane keeps the larger one iff keep_max. One 64-bit compare, the
    // decision is a mask operation on the scalar unit (keys are distinct, or equal zeros where either outcome is the same), two selects.
    auto cx = [](u64 &e, const u64 p, const bool keep_max) {
        const bool lt = e < p;
        e = (lt == keep_max) ? p : e;
    };
    // Merge sizes and distances are compile-time constants in every copy of the body: the lane masks ((lane & j) != 0, the block
    // direction for k <= 64) fold into scalar constants, and for k >= 128 the direction is wave-uniform. (The rolled version spent
    // ~30 VALU instructions per sub-stage on them — 17.5 us for 2048 keys with four waves per SIMD issuing the same code.)
#pragma unroll
    for (int lk = 1; lk <= 11; ++lk) {
        const int k = 1 << lk;
        if (k > n) break;
        if (k >= 256) {
            // distances >= 128: through LDS, one barrier pair per sub-stage
            if (active) { keys[base + lane] = e0; keys[base + 64 + lane] = e1; }
            __syncthreads();
            for (int j = k >> 1; j >= 128; j >>= 1) {
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
        if (!active) continue;
        // the block of merge size k that holds this wave's keys sorts descending iff its index is even
        const bool d0 = k >= 128 ? ((base & k) == 0) : (k == 64 ? true : ((lane & k) == 0));
        const bool d1 = k >= 128 ? ((base & k) == 0) : (k == 64 ? false : ((lane & k) == 0));
#pragma unroll
        for (int lj = 6; lj >= 0; --lj) {
            const int j = 1 << lj;
            if (j > (k >> 1)) continue;
            if (j == 64) {      // partner inside the lane (k >= 128: d0 == d1)
                const bool sw = d0 ? (e0 < e1) : (e0 > e1);
                const u64 a = sw ? e1 : e0, b = sw ? e0 : e1;
                e0 = a; e1 = b;
            } else {
                const u64 p0 = shfl_xor_u64(e0, j), p1 = shfl_xor_u64(e1, j);
                const bool upper = (lane & j) != 0;           // this lane holds the element with the larger index of the pair
                cx(e0, p0, d0 != upper);                      // descending block: the lower index keeps the larger key
                cx(e1, p1, d1 != upper);
            }
        }
    }
    if (active) { keys[base + lane] = e0; keys[base + 64 + lane] = e1; }      // a wave's own 128 slots: nobody else reads them before the barrier
    __syncthreads();
}

__device__ __forceinline__ float sigmoidf_dev(float x) { return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x))); }

// ------------------------------------------------------------------ stage 1: per-class top-K (radix select + sort)
// grid = num_classes, block = 1024. Writes K composite keys (score bits << 32 | ~index) per class, sorted descending;
// slots beyond the number of pixels hold 0.
// Every stage takes up to AQ3D_MAX_HEADS independent problems per launch (blockIdx.y / .z = head): the decode -> NMS -> gather
// chain of a head is a string of single-workgroup, latency-bound launches (~170 us per head), and the 6-head models (nuScenes
// CenterPoint, VoxelNeXt) ran it six times back to back.
#define AQ3D_MAX_HEADS 8
// Rows [row0, row0 + rows) of a sparse head that one decode works on: every live row (sample < 0), or the rows of ONE sample of a
// batch (voxelnext_head.py:418-488 decodes `for k in range(batch_size)` over spatial_indices[:, 0] == k). The rows of a raster-ordered
// level are sorted by sample, so a sample is a contiguous range found by two binary searches; candidate indices are relative to row0
// (the order, and with it every tie, is that of the sample run on its own) and decode_body adds row0 back.
__device__ __forceinline__ void head_rows(const int *__restrict__ coords, const int *__restrict__ count, int cap, int sample, int &row0, int &rows) {
    int n = *count;
    n = n < 0 ? 0 : (n > cap ? cap : n);
    row0 = 0;
    rows = n;
    if (sample < 0 || !coords) return;
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (coords[(size_t)mid * 4] < sample) lo = mid + 1; else hi = mid; }
    row0 = lo;
    hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (coords[(size_t)mid * 4] <= sample) lo = mid + 1; else hi = mid; }
    rows = lo - row0;
}
__device__ __forceinline__ void topk_select_tail(const float *__restrict__ col, int hm_ld, int L, bool filtered, int cache_keys, int K, int Kpad,
                                                 u64 *sel, int *hist, int *sh, unsigned int *keys_s, unsigned int *idx_s,
                                                 u64 *__restrict__ out, const int cls);
__device__ __forceinline__ void topk_per_class_body(const float *__restrict__ hm, int hm_ld, int L_static,
                                                    const int *__restrict__ count, int K, int Kpad,
                                                    int cache_keys, float score_thresh, u64 *__restrict__ out, const int cls) {
    int L = L_static;            // dense map: H*W pixels; sparse head: the live rows (device count, capped by L_static)
    if (count) { L = *count; if (L > L_static) L = L_static; if (L < 0) L = 0; }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *sel = reinterpret_cast<u64 *>(smem_raw);          // [Kpad]
    int *hist = reinterpret_cast<int *>(sel + Kpad);       // [256]
    int *sh = hist + 256;                                  // [16]
    unsigned int *keys_s = reinterpret_cast<unsigned int *>(sh + 16);   // [L] score bit patterns (when they fit in LDS)
    const float *col = hm + cls;
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
        } else {
            // too many scores above the threshold (a heat map that hovers around it): the keys just computed ARE the key cache of
            // the unfiltered path — store them instead of loading and evaluating every pixel a second time
#pragma unroll
            for (int it = 0; it < FPER; ++it) {
                const int i = beg + it * 64 + lane;
                if (it * 64 < chunk && i < end) keys_s[i] = kv[it];
            }
            if (threadIdx.x == 0) s_filtered = 2;
        }
        __syncthreads();
    }
    const bool filtered = try_filter && s_filtered == 1;
    const bool keys_cached = try_filter && s_filtered == 2;
    if (cache_keys && !filtered && !keys_cached) {
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

    topk_select_tail(col, hm_ld, L, filtered, cache_keys, K, Kpad, sel, hist, sh, keys_s, idx_s, out, cls);
}

// Radix select + tie rule + write-out over L keys: keys_s[0..L) when cache_keys (filtered: idx_s[i] = pixel of candidate i, candidates
// in ascending pixel order), else recomputed from the map. Every thread of the 1024 calls it; a barrier separates the caller's LDS
// writes from the first pass.
__device__ __forceinline__ void topk_select_tail(const float *__restrict__ col, int hm_ld, int L, bool filtered, int cache_keys, int K, int Kpad,
                                                 u64 *sel, int *hist, int *sh, unsigned int *keys_s, unsigned int *idx_s,
                                                 u64 *__restrict__ out, const int cls) {
    constexpr int TIE_CAP = 512;
    __shared__ unsigned int s_prefix;
    __shared__ int s_remaining, s_fill, s_ties;
    __shared__ unsigned int tie_idx[TIE_CAP];
    auto key_of = [&](int i) -> unsigned int {
        return cache_keys ? keys_s[i] : __float_as_uint(sigmoidf_dev(col[(size_t)i * hm_ld]));
    };
    auto orig = [&](int i) -> unsigned int { return filtered ? idx_s[i] : (unsigned int)i; };
    const int Keff = K < L ? K : L;
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
            // (measured: the same aggregation on the lower bytes, two rounds then plain atomics, is slower — 41 -> 52 us at 188 x 188)
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

struct TopkArgs {
    const float *hm;
    int hm_ld, L_static;
    const int *count, *coords;
    int sample;
    int K, Kpad, cache_keys, nclass;
    float score_thresh;
    u64 *out;
};
struct TopkBatch {
    TopkArgs p[AQ3D_MAX_HEADS];
};
__global__ __launch_bounds__(1024) void topk_per_class_kernel(TopkBatch b) {
    const TopkArgs &a = b.p[blockIdx.y];
    if ((int)blockIdx.x >= a.nclass) return;
    if (a.count && a.sample >= 0) {       // one sample of a batch: its row range as a "dense map" of that many rows
        int row0, rows;
        head_rows(a.coords, a.count, a.L_static, a.sample, row0, rows);
        topk_per_class_body(a.hm + (size_t)row0 * a.hm_ld, a.hm_ld, rows, nullptr, a.K, a.Kpad, a.cache_keys, a.score_thresh, a.out, (int)blockIdx.x);
        return;
    }
    topk_per_class_body(a.hm, a.hm_ld, a.L_static, a.count, a.K, a.Kpad, a.cache_keys, a.score_thresh, a.out, (int)blockIdx.x);
}

// Dense maps, two launches instead of one workgroup per class. A class column of a [H*W, hm_ld] map is one float per 64-byte line:
// a single workgroup pulls the whole map (3.4 MB at 188 x 188 x 24) through ONE compute unit's texture path and evaluates 36
// sigmoids per thread before the select can start, and the select then walks all H*W keys five times on four SIMDs.
//   topk_scan_kernel    grid (G, classes, heads): workgroup g owns pixels [g*chunk, (g+1)*chunk) (chunk <= 4096, a multiple of 1024),
//                       wave w of it the contiguous 1/16th; scores -> key bits to keys_all[class][pixel]; keys above the score
//                       threshold -> order-preserving compaction (wave counts, 16-wave prefix, ballot ranks) into the chunk's own
//                       candidate range cand[class][g*chunk ..] (composites score << 32 | ~pixel), chunk_cnt[class][g] = how many, and the chunk's histogram
//                       of the candidates over TK_NB bins of the key's top bits (exponent + 5 mantissa bits: 32 bins per octave).
//   topk_select_kernel  grid (classes, heads): sums the chunk histograms, finds the bin that holds the K-th score, and in ONE pass over
//                       the candidates takes every key of a higher bin and collects the pivot bin's keys (normally a few hundred); a
//                       bitonic sort of those as (score, ~pixel) composites yields the rest with the tie rule built in. More than
//                       TK_SORT keys in the pivot bin (saturated or constant maps): the general select over keys_all, as before.
// No cross-workgroup hand-off inside a launch: the kernel boundary is the synchronisation.
struct ScanArgs {
    const float *hm;
    const int *count;                                   // sparse head: live rows on the device (<= L); NULL for a dense map
    const int *coords;                                  // sparse head: its rows' (b, z, y, x)
    int sample;                                         // sparse head: -1 = every row, else the rows of this sample only (head_rows)
    int hm_ld, L, nclass, G, chunk, K, Kpad;            // L: H*W, or the row capacity of a sparse head
    float score_thresh;
    unsigned int *keys_all;                             // [nclass][L]
    u64 *cand;                                          // [nclass][G*chunk] composites: score bits << 32 | ~pixel
    int *chunk_cnt, *chunk_hist;                        // [nclass][G], [nclass][G][TK_NB]
    u64 *out;
};
struct ScanBatch {
    ScanArgs p[AQ3D_MAX_HEADS];
};
constexpr int TK_NB = 4096, TK_SH = 18, TK_SORT = 2048, TK_MAXG = 16;
__device__ __forceinline__ int tk_bin(unsigned int key) {       // monotone in the key; sigmoid <= 1.0 -> bin <= 4064, NaN patterns clamp to the top
    const unsigned int b = key >> TK_SH;
    return b > (unsigned int)(TK_NB - 1) ? TK_NB - 1 : (int)b;
}
__global__ __launch_bounds__(1024) void topk_scan_kernel(ScanBatch b) {
    const ScanArgs &a = b.p[blockIdx.z];
    const int g = (int)blockIdx.x, cls = (int)blockIdx.y;
    if (cls >= a.nclass || g >= a.G) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per_wave = a.chunk >> 4;                       // multiple of 64
    int L = a.L, row0 = 0;
    if (a.count) head_rows(a.coords, a.count, a.L, a.sample, row0, L);
    const int beg = g * a.chunk + w * per_wave, end = min(L, beg + per_wave);
    const float *col = a.hm + cls + (size_t)row0 * a.hm_ld;
    const bool all_pass = a.score_thresh < 0.f;
    const unsigned int tkey = __float_as_uint(a.score_thresh);
    __shared__ int s_wcnt[16];
    __shared__ int s_hist[TK_NB];
    unsigned int kv[4];
    float v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = beg + it * 64 + lane;
        v[it] = (it * 64 < per_wave && i < end) ? col[(size_t)i * a.hm_ld] : 0.f;
    }
    for (int i = threadIdx.x; i < TK_NB; i += 1024) s_hist[i] = 0;
    __syncthreads();
    int cnt = 0;
    u64 pm[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = beg + it * 64 + lane;
        const bool in = it * 64 < per_wave && i < end;
        kv[it] = in ? __float_as_uint(sigmoidf_dev(v[it])) : 0u;
        if (in) a.keys_all[(size_t)cls * a.L + i] = kv[it];
        const bool pass_ = in && (all_pass || kv[it] > tkey);
        pm[it] = __ballot(pass_);
        if (pass_) atomicAdd(&s_hist[tk_bin(kv[it])], 1);
        cnt += (int)__popcll(pm[it]);
    }
    if (lane == 0) s_wcnt[w] = cnt;
    __syncthreads();
    int base = 0, total = 0;
    for (int q = 0; q < 16; ++q) { base += q < w ? s_wcnt[q] : 0; total += s_wcnt[q]; }
    const size_t cbase = ((size_t)cls * a.G + g) * a.chunk;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if ((pm[it] >> lane) & 1ull) {
            const int pos = base + (int)__popcll(pm[it] & ((1ull << lane) - 1ull));
            a.cand[cbase + pos] = ((u64)kv[it] << 32) | (u64)(0xffffffffu - (unsigned int)(beg + it * 64 + lane));
        }
        base += (int)__popcll(pm[it]);
    }
    if (threadIdx.x == 0) a.chunk_cnt[cls * a.G + g] = total;
    int *hdst = a.chunk_hist + ((size_t)cls * a.G + g) * TK_NB;
    for (int i = threadIdx.x; i < TK_NB; i += 1024) hdst[i] = s_hist[i];
}

__global__ __launch_bounds__(1024) void topk_select_kernel(ScanBatch b) {
    const ScanArgs &a = b.p[blockIdx.y];
    const int cls = (int)blockIdx.x;
    if (cls >= a.nclass) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *sel = reinterpret_cast<u64 *>(smem_raw);          // [Kpad]
    int *hist = reinterpret_cast<int *>(sel + a.Kpad);     // [TK_NB]
    u64 *sortbuf = reinterpret_cast<u64 *>(hist + TK_NB);  // [TK_SORT]
    // the general select (crowded pivot bin) needs hist[256] + sh[16] and all L keys: they take over the space from hist + 272 on
    unsigned int *keys_s = reinterpret_cast<unsigned int *>(hist + 272);
    unsigned int *idx_s = keys_s;                           // unused by the unfiltered select
    __shared__ int s_base[TK_MAXG + 1], s_piv, s_rem, s_fill, s_surv, s_sh[16];
    const int tid = threadIdx.x, G = a.G, K = a.K;
    DN_STAMP(0, 0);
    {       // exclusive prefix of the chunk counts: lanes 0..16 of wave 0, one load each
        const int lane = tid & 63;
        if (tid < 64) {
            const int c = lane < G ? a.chunk_cnt[cls * G + lane] : 0;
            int incl = c;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            if (lane <= TK_MAXG) s_base[lane] = incl - c;
        }
        if (tid == 0) { s_piv = -1; s_rem = 0; s_fill = 0; s_surv = 0; }
    }
    // 1. histogram of all candidates of the class
    {
        int h[TK_NB / 1024];
#pragma unroll
        for (int u = 0; u < TK_NB / 1024; ++u) h[u] = 0;
        const int *src = a.chunk_hist + (size_t)cls * G * TK_NB;
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int u = 0; u < TK_NB / 1024; ++u) h[u] += src[(size_t)g * TK_NB + u * 1024 + tid];
#pragma unroll
        for (int u = 0; u < TK_NB / 1024; ++u) hist[u * 1024 + tid] = h[u];
    }
    for (int i = tid; i < a.Kpad; i += 1024) sel[i] = 0ull;
    __syncthreads();
    DN_STAMP(0, 1);
    const int n = s_base[TK_MAXG];
    const int Keff = K < n ? K : n;
    if (Keff == 0) {        // nothing above the threshold
        for (int i = tid; i < K; i += 1024) a.out[(size_t)cls * K + i] = 0ull;
        return;
    }
    // 2. the bin of the Keff-th largest key: thread t owns the bins TK_NB-1-4t .. TK_NB-4-4t (descending), block prefix from the top
    {
        constexpr int PER = TK_NB / 1024;
        int h[PER], local = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) { h[u] = hist[TK_NB - 1 - (tid * PER + u)]; local += h[u]; }
        int tot;
        int above = block_scan_excl_1024(local, s_sh, &tot);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (above < Keff && above + h[u] >= Keff) { s_piv = TK_NB - 1 - (tid * PER + u); s_rem = Keff - above; }     // exactly one (thread, u)
            above += h[u];
        }
    }
    __syncthreads();
    const int piv = s_piv, rem = s_rem, m = hist[piv], pre = Keff - rem;
    DN_STAMP(0, 2);
    if (m > TK_SORT) {
        // crowded pivot bin: the general select over every pixel's key (keys_all -> LDS), ties by pixel order
        __syncthreads();        // hist[piv] read by everyone before the select reuses hist
        int L = a.L, row0 = 0;
        if (a.count) head_rows(a.coords, a.count, a.L, a.sample, row0, L);
        const unsigned int *src = a.keys_all + (size_t)cls * a.L;
        for (int base = tid; base < L; base += 8 * 1024) {
            unsigned int t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = base + u * 1024 < L ? src[base + u * 1024] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (base + u * 1024 < L) keys_s[base + u * 1024] = t[u];
        }
        topk_select_tail(a.hm + cls + (size_t)row0 * a.hm_ld, a.hm_ld, L, false, 1, K, a.Kpad, sel, hist, hist + 256, keys_s, idx_s, a.out, cls);
        return;
    }
    // 3. one pass over the candidates: higher bin -> selected, pivot bin -> composite into the sort buffer
    for (int i = tid; i < TK_SORT; i += 1024) sortbuf[i] = 0ull;
    __syncthreads();
    DN_STAMP(0, 3);
    // chunk-major, three chunks (12 eight-byte loads per thread) per round trip; the chunk counts sit in registers (static indices)
    {
        int cntg[TK_MAXG];
#pragma unroll
        for (int g = 0; g < TK_MAXG; ++g) cntg[g] = s_base[g + 1] - s_base[g];
#pragma unroll
        for (int r = 0; r < (TK_MAXG + 2) / 3; ++r) {
            if (r * 3 >= G) break;
            u64 cc[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int g = r * 3 + u / 4, j = tid + (u & 3) * 1024;
                cc[u] = (g < TK_MAXG && j < cntg[g < TK_MAXG ? g : 0]) ? a.cand[((size_t)cls * G + g) * a.chunk + j] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                if (cc[u] == 0ull) continue;        // a candidate's composite is never 0 (its low word is ~pixel)
                const int bin = tk_bin((unsigned int)(cc[u] >> 32));
                if (bin > piv) sel[atomicAdd(&s_fill, 1)] = cc[u];
                else if (bin == piv) sortbuf[atomicAdd(&s_surv, 1)] = cc[u];
            }
        }
    }
    __syncthreads();
    DN_STAMP(0, 4);
    // 4. the pivot bin's keys, best first ((score, ~pixel) composites: equal scores order by ascending pixel): the first `rem` go in
    if (m <= 512) {         // the usual case: rank by counting (m broadcast reads per key) instead of a 45-stage bitonic network
        if (tid < m) {
            const u64 mine = sortbuf[tid];
            int rank = 0;
            for (int j0 = 0; j0 < m; j0 += 8) {         // slots beyond m hold 0 (never greater); 8 broadcast reads in flight
                u64 o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = sortbuf[j0 + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += o[u] > mine ? 1 : 0;      // composites are distinct
            }
            if (rank < rem) sel[pre + rank] = mine;
        }
    } else {
        int npad = 1024;
        while (npad < m) npad <<= 1;
        bitonic_sort_desc_hybrid(sortbuf, npad);
        for (int i = tid; i < rem; i += 1024) sel[pre + i] = sortbuf[i];
    }
    __syncthreads();
    DN_STAMP(0, 5);
    for (int i = tid; i < K; i += 1024) a.out[(size_t)cls * K + i] = (i < Keff) ? sel[i] : 0ull;
    DN_STAMP(0, 6);
}

// ------------------------------------------------------------------ stage 2: top-K over classes*K, decode, mask
struct DecodeParams {
    aq3d_decode_cfg c;
    const float *center, *center_z, *dim, *rot, *vel;
    const int *coords;      // sparse head: rows (b, z|0, y, x) of the voxels; NULL for a dense map
    const int *count;       // sparse head: live rows
    int cap, sample;        // sparse head: row capacity; -1 or the sample whose rows were ranked (candidate indices are relative to its first row)
};

__device__ __forceinline__ void decode_body(const u64 *__restrict__ cand, int ncand, int Npad, const DecodeParams &q,
                                            float *__restrict__ boxes_out, float *__restrict__ scores_out,
                                            int *__restrict__ labels_out, int *__restrict__ count_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *keys = reinterpret_cast<u64 *>(smem_raw);  // [Npad] : score bits << 32 | ~position (position = class*K + rank)
    __shared__ int sh[16];
    const int K = q.c.K;
    int row0 = 0;
    if (q.coords && q.sample >= 0) { int rows_; head_rows(q.coords, q.count, q.cap, q.sample, row0, rows_); }
    DN_STAMP(1, 0);
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
    DN_STAMP(1, 1);
    if (Npad >= 128 && Npad <= 2048 && blockDim.x == 1024) bitonic_sort_desc_hybrid(keys, Npad);
    else bitonic_sort_desc(keys, Npad);
    DN_STAMP(1, 2);

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
        const int pix = (int)(cp & 0x07ffffffu) + row0;
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
    DN_STAMP(1, 3);
    int total;
    const int dst = block_scan_excl_1024(valid, sh, &total);
    DN_STAMP(1, 4);
    if (valid) {
        for (int j = 0; j < nbox; ++j) boxes_out[(size_t)dst * nbox + j] = box[j];
        scores_out[dst] = score;
        labels_out[dst] = label;
    }
    if (threadIdx.x == 0) *count_out = total;
    DN_STAMP(1, 5);
}

struct DecodeArgs {
    const u64 *cand;
    int ncand, Npad;
    DecodeParams q;
    float *boxes_out, *scores_out;
    int *labels_out, *count_out;
};
struct DecodeBatch {
    DecodeArgs p[AQ3D_MAX_HEADS];
};
__global__ __launch_bounds__(1024) void decode_kernel(DecodeBatch b) {
    const DecodeArgs &a = b.p[blockIdx.x];
    decode_body(a.cand, a.ncand, a.Npad, a.q, a.boxes_out, a.scores_out, a.labels_out, a.count_out);
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// workspace: [classes][K] selected keys | (dense maps whose keys fit LDS) keys_all | cand | chunk_cnt | chunk_hist
struct TopkWs { size_t sel, keys_all, cand, chunk_cnt, chunk_hist; int G, chunk, multi; };
static size_t topk_ws_layout(const aq3d_decode_cfg *cfg, long L, TopkWs *w) {     // L: pixels of the map / row capacity of the sparse head
    TopkWs t;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    t.sel = take((size_t)cfg->num_classes * cfg->K * sizeof(u64) + 256);
    t.multi = (L > 0 && L * 4 <= 141 * 1024) ? 1 : 0;          // the same bound as cache_keys below
    t.G = t.chunk = 0;
    t.keys_all = t.cand = t.chunk_cnt = t.chunk_hist = 0;
    if (t.multi) {
        t.G = (int)((L + 4095) / 4096);
        t.chunk = (int)(((L + t.G - 1) / t.G + 1023) / 1024 * 1024);
        t.keys_all = take((size_t)cfg->num_classes * L * 4);
        t.cand = take((size_t)cfg->num_classes * t.G * t.chunk * 8);
        t.chunk_cnt = take((size_t)cfg->num_classes * t.G * 4);
        t.chunk_hist = take((size_t)cfg->num_classes * t.G * TK_NB * 4);
    }
    if (w) *w = t;
    return off;
}

extern "C" size_t aq3d_center_decode_workspace_bytes(const aq3d_decode_cfg *cfg) {
    if (!cfg) return 0;
    return topk_ws_layout(cfg, (long)cfg->H * cfg->W, nullptr);
}

extern "C" size_t aq3d_voxel_decode_workspace_bytes(const aq3d_decode_cfg *cfg, int32_t cap) {
    if (!cfg) return 0;
    return topk_ws_layout(cfg, cap, nullptr);
}
static size_t decode_ws_min(const aq3d_decode_cfg *cfg) { return (size_t)cfg->num_classes * cfg->K * sizeof(u64) + 256; }

// One decode problem: dense maps (coords == NULL: aq3d_center_decode) or rows of a sparse head (aq3d_voxel_decode).
struct DecodeProblem {
    const float *hm, *center, *center_z, *dim, *rot, *vel;
    const int32_t *coords, *count;
    int32_t cap, sample;
    const aq3d_decode_cfg *cfg;
    float *boxes_out, *scores_out;
    int32_t *labels_out, *count_out;
    void *workspace;
    size_t workspace_bytes;
};

static int check_decode(const DecodeProblem &d, const char *who) {
    AQ3D_REQUIRE(d.hm && d.center && d.center_z && d.dim && d.rot && d.cfg && d.boxes_out && d.scores_out && d.labels_out && d.count_out &&
                     d.workspace, "%s: null argument", who);
    AQ3D_REQUIRE(!d.cfg->has_vel || d.vel, "%s: has_vel without vel", who);
    AQ3D_REQUIRE(d.cfg->num_classes >= 1 && d.cfg->num_classes <= 16, "%s: num_classes=%d", who, d.cfg->num_classes);
    AQ3D_REQUIRE(d.cfg->K >= 1 && d.cfg->K <= 512, "%s: K=%d not in 1..512", who, d.cfg->K);
    if (d.coords) AQ3D_REQUIRE(d.count && d.cap >= 1, "%s: sparse head without count / cap=%d", who, d.cap);
    else AQ3D_REQUIRE(d.cfg->H > 0 && d.cfg->W > 0, "%s: empty map", who);
    if (d.workspace_bytes < decode_ws_min(d.cfg)) {     // the full *_workspace_bytes() size enables the multi-workgroup top-K, this much is required
        aq3d_set_error("%s: workspace too small", who);
        return AQ3D_ERR_WORKSPACE;
    }
    return AQ3D_OK;
}

// top-K per class + decode of n <= AQ3D_MAX_HEADS checked problems: two launches
static void launch_decode(const DecodeProblem *d, int n, hipStream_t stream) {
    TopkBatch tb;
    ScanBatch sb;
    DecodeBatch db;
    size_t lds1_max = 0, lds2_max = 0;
    int max_classes = 1, max_g = 1;
    bool multi = true;          // maps / row sets whose key cache fits LDS and whose workspace has room: scores evaluated by many workgroups
    for (int i = 0; i < n; ++i) {
        TopkWs w;
        const long Ls = d[i].coords ? (long)d[i].cap : (long)d[i].cfg->H * d[i].cfg->W;
        const size_t need = topk_ws_layout(d[i].cfg, Ls, &w);
        multi = multi && w.multi && d[i].workspace_bytes >= need;
        ScanArgs &c = sb.p[i];
        char *ws = (char *)d[i].workspace;
        c.coords = d[i].coords; c.sample = d[i].coords ? d[i].sample : -1;
        c.hm = d[i].hm; c.count = d[i].coords ? d[i].count : nullptr; c.hm_ld = d[i].cfg->hm_ld; c.L = (int)Ls; c.nclass = d[i].cfg->num_classes; c.G = w.G; c.chunk = w.chunk;
        c.K = d[i].cfg->K; c.Kpad = next_pow2(c.K); c.score_thresh = d[i].cfg->score_thresh;
        c.keys_all = (unsigned int *)(ws + w.keys_all); c.cand = (u64 *)(ws + w.cand);
        c.chunk_cnt = (int *)(ws + w.chunk_cnt); c.chunk_hist = (int *)(ws + w.chunk_hist); c.out = (u64 *)(ws + w.sel);
        max_g = w.G > max_g ? w.G : max_g;
    }
    for (int i = 0; i < n; ++i) {
        const aq3d_decode_cfg *cfg = d[i].cfg;
        const int L = d[i].coords ? d[i].cap : cfg->H * cfg->W, K = cfg->K, Kpad = next_pow2(K);
        size_t lds1 = (size_t)Kpad * 8 + 256 * 4 + 16 * 4;
        // key cache [L] — or, in the pre-filtered mode, keys + pixel indices of up to 16384 candidates (2 x 64 KiB); the
        // order-preserving score-threshold pre-filter is always on (identical detections)
        const size_t key_bytes = (size_t)L * 4 > 131072 ? (size_t)L * 4 : 131072;
        const int cache_keys = (lds1 + key_bytes <= 150 * 1024) ? 1 : 0;   // one workgroup may use up to 160 KiB of LDS
        if (cache_keys) lds1 += key_bytes;
        TopkArgs &t = tb.p[i];
        t.hm = d[i].hm; t.hm_ld = cfg->hm_ld; t.L_static = L; t.count = d[i].coords ? d[i].count : nullptr; t.K = K; t.Kpad = Kpad;
        t.coords = d[i].coords; t.sample = d[i].coords ? d[i].sample : -1;
        t.cache_keys = cache_keys; t.nclass = cfg->num_classes; t.score_thresh = cfg->score_thresh; t.out = (u64 *)d[i].workspace;
        DecodeArgs &a = db.p[i];
        a.cand = (const u64 *)d[i].workspace; a.ncand = cfg->num_classes * K; a.Npad = next_pow2(a.ncand);
        a.q.c = *cfg; a.q.center = d[i].center; a.q.center_z = d[i].center_z; a.q.dim = d[i].dim; a.q.rot = d[i].rot; a.q.vel = d[i].vel;
        a.q.coords = d[i].coords; a.q.count = d[i].count; a.q.cap = d[i].cap; a.q.sample = d[i].coords ? d[i].sample : -1;
        a.boxes_out = d[i].boxes_out; a.scores_out = d[i].scores_out; a.labels_out = d[i].labels_out; a.count_out = d[i].count_out;
        lds1_max = lds1 > lds1_max ? lds1 : lds1_max;
        const size_t lds2 = (size_t)a.Npad * 8;
        lds2_max = lds2 > lds2_max ? lds2 : lds2_max;
        max_classes = cfg->num_classes > max_classes ? cfg->num_classes : max_classes;
    }
    for (int i = n; i < AQ3D_MAX_HEADS; ++i) { tb.p[i] = tb.p[0]; db.p[i] = db.p[0]; sb.p[i] = sb.p[0]; }
    if (multi) {
        size_t lds_sel = 0;
        for (int i = 0; i < n; ++i) {
            const size_t a_ = (size_t)TK_NB * 4 + (size_t)TK_SORT * 8, b_ = 272 * 4 + (size_t)sb.p[i].L * 4;
            const size_t l = (size_t)sb.p[i].Kpad * 8 + (a_ > b_ ? a_ : b_);
            lds_sel = l > lds_sel ? l : lds_sel;
        }
        hipLaunchKernelGGL(topk_scan_kernel, dim3(max_g, max_classes, n), dim3(1024), 0, stream, sb);
        if (lds_sel > 64 * 1024)
            hipFuncSetAttribute(reinterpret_cast<const void *>(topk_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sel);
        hipLaunchKernelGGL(topk_select_kernel, dim3(max_classes, n), dim3(1024), lds_sel, stream, sb);
    } else {
        if (lds1_max > 64 * 1024)
            hipFuncSetAttribute(reinterpret_cast<const void *>(topk_per_class_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1_max);
        hipLaunchKernelGGL(topk_per_class_kernel, dim3(max_classes, n), dim3(1024), lds1_max, stream, tb);
    }
    if (lds2_max > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2_max);
    hipLaunchKernelGGL(decode_kernel, dim3(n), dim3(1024), lds2_max, stream, db);
}

extern "C" int aq3d_center_decode(const float *hm, const float *center, const float *center_z, const float *dim,
                                  const float *rot, const float *vel, const aq3d_decode_cfg *cfg, float *boxes_out,
                                  float *scores_out, int32_t *labels_out, int32_t *count_out, void *workspace,
                                  size_t workspace_bytes, void *stream_) {
    const DecodeProblem d = {hm, center, center_z, dim, rot, vel, nullptr, nullptr, 0, -1, cfg, boxes_out, scores_out, labels_out, count_out,
                             workspace, workspace_bytes};
    const int rc = check_decode(d, "aq3d_center_decode");
    if (rc != AQ3D_OK) return rc;
    launch_decode(&d, 1, (hipStream_t)stream_);
    AQ3D_CHECK_LAUNCH("aq3d_center_decode");
    return AQ3D_OK;
}

extern "C" int aq3d_voxel_decode(const float *hm, const float *center, const float *center_z, const float *dim,
                                 const float *rot, const float *vel, const int32_t *coords, const int32_t *count, int32_t cap,
                                 const aq3d_decode_cfg *cfg, float *boxes_out, float *scores_out, int32_t *labels_out,
                                 int32_t *count_out, void *workspace, size_t workspace_bytes, void *stream_) {
    AQ3D_REQUIRE(coords, "aq3d_voxel_decode: null coords");
    const DecodeProblem d = {hm, center, center_z, dim, rot, vel, coords, count, cap, -1, cfg, boxes_out, scores_out, labels_out, count_out,
                             workspace, workspace_bytes};
    const int rc = check_decode(d, "aq3d_voxel_decode");
    if (rc != AQ3D_OK) return rc;
    launch_decode(&d, 1, (hipStream_t)stream_);
    AQ3D_CHECK_LAUNCH("aq3d_voxel_decode");
    return AQ3D_OK;
}

// ------------------------------------------------------------------ rotated NMS (geometry: nms_geom.h)
// one wave per (row box, column block): lane j tests box (cb*64 + j) against the row box, ballot -> one mask word
struct NmsArgs {
    const float *boxes;
    int box_ld;
    const int *count;
    int cap;
    float thresh;
    int col_blocks, post_max, use_lds;
    u64 *mask;
    int *keep, *keep_count;
};
struct NmsBatch {
    NmsArgs p[AQ3D_MAX_HEADS];
};
__global__ __launch_bounds__(64) void nms_mask_kernel(NmsBatch nb) {
    const NmsArgs &q = nb.p[blockIdx.z];
    const float *__restrict__ boxes = q.boxes;
    const int box_ld = q.box_ld, cap = q.cap, col_blocks = q.col_blocks;
    const float thresh = q.thresh;
    u64 *__restrict__ mask = q.mask;
    int n = *q.count;
    if (n > cap) n = cap;
    const int row = blockIdx.x, cb = blockIdx.y;
    if (row >= n || cb >= col_blocks) return;
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
__global__ __launch_bounds__(256) void nms_sweep_kernel(NmsBatch nb_) {
    const NmsArgs &q = nb_.p[blockIdx.x];
    const u64 *__restrict__ mask = q.mask;
    const int *count = q.count;
    const int cap = q.cap, col_blocks = q.col_blocks, post_max = q.post_max, use_lds = q.use_lds;
    int *__restrict__ keep = q.keep, *__restrict__ keep_count = q.keep_count;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *mask_s = reinterpret_cast<u64 *>(smem_raw);
    DN_STAMP(2, 0);
    int n = *count;
    if (n > cap) n = cap;
    const u64 *msrc = mask;
    DN_STAMP(2, 1);
    if (use_lds) {
        const int words = n * col_blocks;
        for (int i0 = threadIdx.x; i0 < words; i0 += 8 * 256) {      // 8 independent loads per thread and round trip (blockDim.x == 256)
            u64 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = i0 + u * 256 < words ? mask[i0 + u * 256] : 0ull;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * 256 < words) mask_s[i0 + u * 256] = t[u];
        }
        __syncthreads();
        msrc = mask_s;
    }
    DN_STAMP(2, 2);
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    __shared__ u64 remv_s[64];          // removal word per column block (cap <= 4096)
    remv_s[lane] = 0ull;
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
        // Greedy pass over the block: box t is kept iff bit t of `cur` is clear when t is visited, and a kept box ORs in its diagonal
        // word (bits above t only). Boxes whose diagonal word is empty never change `cur`, so only the others are visited, in order
        // (s_ff1 over their ballot): after the walk kept = ~cur. (64 unconditional steps cost 2.2 us per block on the scalar unit —
        // ~85 cycles per step for a lone wave — whether or not anything overlapped.)
        u64 cur = remv_s[b];
        cur = bcast(cur, 0);
        u64 todo = __ballot(diag != 0ull);
        while (todo) {
            const int t = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            if (!((cur >> t) & 1ull)) cur |= bcast(diag, t);
        }
        const int lim = min(64, n - b * 64);
        u64 kept = ~cur;
        if (lim < 64) kept &= (1ull << lim) - 1ull;
        if (b == 0) DN_STAMP(2, 3);
        if ((kept >> lane) & 1ull) {
            const int pos = num + __popcll(kept & ((1ull << lane) - 1ull));
            if (pos < post_max) keep[pos] = row;
        }
        num += __popcll(kept);
        if (b == 0) DN_STAMP(2, 4);
        if (b + 1 >= nb) break;          // nothing behind the last block
        // remv[c] |= OR of the kept rows' words of column block c, for c > b: lane = one kept row, LDS OR-atomics into the removal words
        // (the wave's LDS operations complete in order: the next block's read of remv_s sees them)
        if ((kept >> lane) & 1ull) {
            for (int c = b + 1; c < col_blocks; ++c) {
                const u64 mw = msrc[(size_t)row * col_blocks + c];
                if (mw) atomicOr(reinterpret_cast<unsigned long long *>(&remv_s[c]), mw);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (b == 0) DN_STAMP(2, 5);
    }
    if (lane == 0) *keep_count = num < post_max ? num : post_max;
    DN_STAMP(2, 6);
}

extern "C" size_t aq3d_nms_workspace_bytes(int32_t cap) {
    if (cap < 1) cap = 1;
    return (size_t)cap * aq3d_div_up(cap, 64) * sizeof(u64) + 256;
}

struct NmsProblem {
    const float *boxes;
    int32_t box_ld;
    const int32_t *count;
    int32_t cap;
    float thresh;
    int32_t post_max;
    int32_t *keep_out, *keep_count;
    void *workspace;
    size_t workspace_bytes;
};

static int check_nms(const NmsProblem &q, const char *who) {
    AQ3D_REQUIRE(q.boxes && q.count && q.keep_out && q.keep_count && q.workspace, "%s: null argument", who);
    AQ3D_REQUIRE(q.cap >= 1 && q.cap <= 4096 && q.box_ld >= 7 && q.post_max >= 1, "%s: cap=%d box_ld=%d post_max=%d", who, q.cap, q.box_ld,
                 q.post_max);
    if (q.workspace_bytes < aq3d_nms_workspace_bytes(q.cap)) {
        aq3d_set_error("%s: workspace too small", who);
        return AQ3D_ERR_WORKSPACE;
    }
    return AQ3D_OK;
}

// ballot mask + greedy sweep of n <= AQ3D_MAX_HEADS checked problems: two launches
static void launch_nms(const NmsProblem *q, int n, hipStream_t stream) {
    NmsBatch nb;
    int max_cap = 1, max_cb = 1;
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        NmsArgs &a = nb.p[i];
        a.boxes = q[i].boxes; a.box_ld = q[i].box_ld; a.count = q[i].count; a.cap = q[i].cap; a.thresh = q[i].thresh;
        a.col_blocks = aq3d_div_up(q[i].cap, 64); a.post_max = q[i].post_max; a.mask = (u64 *)q[i].workspace;
        a.keep = q[i].keep_out; a.keep_count = q[i].keep_count;
        const size_t mask_bytes = (size_t)q[i].cap * a.col_blocks * sizeof(u64);
        a.use_lds = mask_bytes <= 128 * 1024;
        if (a.use_lds && mask_bytes > lds) lds = mask_bytes;
        max_cap = q[i].cap > max_cap ? q[i].cap : max_cap;
        max_cb = a.col_blocks > max_cb ? a.col_blocks : max_cb;
    }
    for (int i = n; i < AQ3D_MAX_HEADS; ++i) nb.p[i] = nb.p[0];
    hipLaunchKernelGGL(nms_mask_kernel, dim3(max_cap, max_cb, n), dim3(64), 0, stream, nb);
    if (lds > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(nms_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(n), dim3(256), lds, stream, nb);
}

extern "C" int aq3d_nms_rotated(const float *boxes, int32_t box_ld, const int32_t *count, int32_t cap, float thresh,
                                int32_t post_max, int32_t *keep_out, int32_t *keep_count, void *workspace,
                                size_t workspace_bytes, void *stream_) {
    const NmsProblem q = {boxes, box_ld, count, cap, thresh, post_max, keep_out, keep_count, workspace, workspace_bytes};
    const int rc = check_nms(q, "aq3d_nms_rotated");
    if (rc != AQ3D_OK) return rc;
    launch_nms(&q, 1, (hipStream_t)stream_);
    AQ3D_CHECK_LAUNCH("aq3d_nms_rotated");
    return AQ3D_OK;
}

struct GatherArgs {
    const float *boxes;
    int box_ld;
    const float *scores;
    const int *labels, *class_map, *keep, *keep_count;
    int cap;
    float *boxes_out, *scores_out;
    long long *labels_out;
};
struct GatherBatch {
    GatherArgs p[AQ3D_MAX_HEADS];
};
__global__ __launch_bounds__(256) void gather_dets_kernel(GatherBatch gb) {
    const GatherArgs &a = gb.p[blockIdx.y];
    int n = *a.keep_count;
    if (n > a.cap) n = a.cap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int src = a.keep[i];
        for (int j = 0; j < a.box_ld; ++j) a.boxes_out[(size_t)i * a.box_ld + j] = a.boxes[(size_t)src * a.box_ld + j];
        a.scores_out[i] = a.scores[src];
        const int lab = a.labels[src];
        a.labels_out[i] = (long long)(a.class_map ? a.class_map[lab] : lab) + 1;  // center_head.py:363
    }
}

static void launch_gather(const GatherArgs *g, int n, hipStream_t stream) {
    GatherBatch gb;
    int max_cap = 1;
    for (int i = 0; i < n; ++i) { gb.p[i] = g[i]; max_cap = g[i].cap > max_cap ? g[i].cap : max_cap; }
    for (int i = n; i < AQ3D_MAX_HEADS; ++i) gb.p[i] = gb.p[0];
    hipLaunchKernelGGL(gather_dets_kernel, dim3(aq3d_grid(max_cap, 256), n), dim3(256), 0, stream, gb);
}

extern "C" int aq3d_gather_dets(const float *boxes, int32_t box_ld, const float *scores, const int32_t *labels,
                                const int32_t *class_map, const int32_t *keep, const int32_t *keep_count, int32_t cap,
                                float *boxes_out, float *scores_out, int64_t *labels_out, void *stream_) {
    AQ3D_REQUIRE(boxes && scores && labels && keep && keep_count && boxes_out && scores_out && labels_out,
                 "aq3d_gather_dets: null argument");
    const GatherArgs g = {boxes, box_ld, scores, labels, class_map, keep, keep_count, cap, boxes_out, scores_out, (long long *)labels_out};
    launch_gather(&g, 1, (hipStream_t)stream_);
    AQ3D_CHECK_LAUNCH("aq3d_gather_dets");
    return AQ3D_OK;
}

// The whole post-processing chain of up to AQ3D_MAX_HEADS heads in FIVE launches (top-K, decode, mask, sweep, gather):
// center_head.py:297-365 / voxelnext_head.py:418-560 run it head after head.
extern "C" int aq3d_decode_nms_batch(const aq3d_head_problem *heads, int32_t n, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(heads && n >= 1 && n <= AQ3D_MAX_HEADS, "aq3d_decode_nms_batch: n=%d not in 1..%d", n, AQ3D_MAX_HEADS);
    DecodeProblem d[AQ3D_MAX_HEADS];
    NmsProblem q[AQ3D_MAX_HEADS];
    GatherArgs g[AQ3D_MAX_HEADS];
    for (int i = 0; i < n; ++i) {
        const aq3d_head_problem &h = heads[i];
        AQ3D_REQUIRE(h.cfg, "aq3d_decode_nms_batch: head %d: null cfg", i);
        const int nbox = h.cfg->has_vel ? 9 : 7;
        AQ3D_REQUIRE(h.sample >= -1, "aq3d_decode_nms_batch: head %d: sample=%d", i, h.sample);
        d[i] = {h.hm, h.center, h.center_z, h.dim, h.rot, h.vel, h.coords, h.count, h.cap, h.sample, h.cfg, h.cand_boxes, h.cand_scores, h.cand_labels,
                h.cand_count, h.decode_workspace, h.decode_workspace_bytes};
        int rc = check_decode(d[i], "aq3d_decode_nms_batch(decode)");
        if (rc != AQ3D_OK) return rc;
        AQ3D_REQUIRE(h.nms_cap >= 1 && h.nms_cap <= h.cfg->K, "aq3d_decode_nms_batch: head %d: nms_cap=%d K=%d", i, h.nms_cap, h.cfg->K);
        q[i] = {h.cand_boxes, nbox, h.cand_count, h.nms_cap, h.nms_thresh, h.post_max, h.keep, h.keep_count, h.nms_workspace, h.nms_workspace_bytes};
        rc = check_nms(q[i], "aq3d_decode_nms_batch(nms)");
        if (rc != AQ3D_OK) return rc;
        AQ3D_REQUIRE(h.boxes_out && h.scores_out && h.labels_out, "aq3d_decode_nms_batch: head %d: null output", i);
        g[i] = {h.cand_boxes, nbox, h.cand_scores, h.cand_labels, h.class_map, h.keep, h.keep_count, h.nms_cap, h.boxes_out, h.scores_out,
                (long long *)h.labels_out};
    }
    launch_decode(d, n, stream);
    launch_nms(q, n, stream);
    launch_gather(g, n, stream);
    AQ3D_CHECK_LAUNCH("aq3d_decode_nms_batch");
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
