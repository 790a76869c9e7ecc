// Sparse index levels: which sites are active, in which row each one lives, and the gather tables of the sparse convs.
//
// The reference gets all of this from spconv (SubMConv3d / SparseConv3d "indice pairs", call sites
// pcdet/models/backbones_3d/spconv_backbone.py:12-15,39-46,194-234; voxel generator: pcdet/datasets/processor/
// data_processor.py:45-61). Restated semantics:
//   out[o] = sum_k W[k] * in[o*stride - pad + k]      (k = (kz,ky,kx) row-major, cross-correlation)
//   SubM: active outputs == active inputs (stride 1, pad = k/2).
//   SparseConv: o is active iff at least one active input falls in its window; out_shape = (in+2p-k)/s + 1.
//
// MI355X layout (round 3; rounds 1-2 used 64-bit-key hash tables, per-candidate atomics and three-launch scans):
//   * A level is a BITMAP over its grid, one 64-bit word per 8x8 (y, x) patch of one z slice, plus the number of sites in
//     the words before each word. site -> rank is two loads (word, base) and a popcount: no probing, and the 27 neighbours
//     of a site touch 3-12 words that its row neighbours touch too.
//   * The active set of a strided conv is a MORPHOLOGICAL operation on that bitmap: one thread per output word ORs shifted /
//     compressed input rows with 32-bit bit tricks. No candidates, no atomics, no memset, deterministic by construction.
//   * Rows are numbered by one stable counting pass over the raster rank, keyed by which z / y / x planes of the 3x3x3
//     neighbourhood hold any neighbour (<= 128 classes, heaviest class first). A 128-row tile of the gather-GEMM kernels then
//     executes 0.21 / 0.44 / 0.52 / 0.68 of its 27 kernel offsets on the four levels of a W150k frame instead of 1.0 (a full
//     sort by the 27-bit mask gives 0.16 / 0.42 / 0.49 / 0.62; scratch/order_study.py), and the heavy tiles are dispatched first.
//   * Gather tables stay OUTPUT-stationary: nbr[out_row][k] = in_row | -1, one coalesced i32 row per output site.
// Row order is not observable through the detector (sums run over k in a fixed order per output row; reductions are maxima).
#include "common.h"

#define IX_BLOCK 256

struct Lv {
    int Z, Y, X, B, cap, PY, PX, words;
    int sorted;         // rows numbered class by class (AQ3D_LEVEL_SORTED) or in bitmap raster order
    unsigned long long *bits;
    int *base, *rank2row, *coords, *count, *total;
};

struct IxWs {
    int *ticket;        // [16] zero between launches (self-resetting)
    int *sstart;        // [batch + 1] rank of the first site of every sample
    int *cls_tot;       // [128] sites per class
    int *chunk_sums;    // [words / 256 + 1]
    int *chunk_flag;    // [words / 256 + 1] level-0 builders: 1 = some point / voxel marked a word of the chunk (everything else skips the empty ones)
    int *cell_of_rank;  // [cap]
    int *hist;          // [(cap / 256 + 1) * 128]
    int *pcell;         // [points]
    int *next;          // [points]
    int *head;          // [cap]
    unsigned char *key_of_rank;   // [cap]
};

static size_t ix_align(size_t v) { return (v + 255) / 256 * 256; }
#define IX_MAX_BATCH 64

extern "C" int64_t aq3d_level_words(const int32_t shape[3], int32_t batch) {
    return (int64_t)batch * shape[0] * ((shape[1] + 7) / 8) * ((shape[2] + 7) / 8);
}

extern "C" size_t aq3d_index_workspace_bytes(int32_t max_cap, int64_t max_words, int32_t max_points) {
    size_t b = 256 + ix_align((IX_MAX_BATCH + 1) * 4) + 512;
    b += 2 * ix_align(((size_t)max_words / IX_BLOCK + 2) * 4);
    b += ix_align((size_t)max_cap * 4);
    b += ix_align(((size_t)max_cap / IX_BLOCK + 2) * 128 * 4);
    b += 2 * ix_align((size_t)(max_points > 0 ? max_points : 1) * 4);
    b += ix_align((size_t)max_cap * 4);
    b += ix_align((size_t)max_cap);
    return b;
}

static bool ix_carve(void *ws, size_t bytes, int cap, int64_t words, int points, IxWs *w) {
    if (!ws || bytes < aq3d_index_workspace_bytes(cap, words, points)) return false;
    char *p = (char *)ws;
    w->ticket = (int *)p;       p += 256;
    w->sstart = (int *)p;       p += ix_align((IX_MAX_BATCH + 1) * 4);
    w->cls_tot = (int *)p;      p += 512;
    w->chunk_sums = (int *)p;   p += ix_align(((size_t)words / IX_BLOCK + 2) * 4);
    w->chunk_flag = (int *)p;   p += ix_align(((size_t)words / IX_BLOCK + 2) * 4);
    w->cell_of_rank = (int *)p; p += ix_align((size_t)cap * 4);
    w->hist = (int *)p;         p += ix_align(((size_t)cap / IX_BLOCK + 2) * 128 * 4);
    w->pcell = (int *)p;        p += ix_align((size_t)(points > 0 ? points : 1) * 4);
    w->next = (int *)p;         p += ix_align((size_t)(points > 0 ? points : 1) * 4);
    w->head = (int *)p;         p += ix_align((size_t)cap * 4);
    w->key_of_rank = (unsigned char *)p;
    return true;
}

static int lv_of(const aq3d_level *l, Lv *o, const char *who) {
    AQ3D_REQUIRE(l && l->bits && l->base && l->row_of_rank && l->coords && l->count && l->total, "%s: null level field", who);
    AQ3D_REQUIRE(l->shape[0] > 0 && l->shape[1] > 0 && l->shape[2] > 0 && l->batch > 0 && l->batch <= IX_MAX_BATCH && l->cap > 0,
                 "%s: bad level geometry", who);
    const int64_t words = aq3d_level_words(l->shape, l->batch);
    AQ3D_REQUIRE(words * 64 < (int64_t)0x7fffffff, "%s: grid of %lld cells overflows 32-bit cell ids", who, (long long)words * 64);
    o->Z = l->shape[0]; o->Y = l->shape[1]; o->X = l->shape[2]; o->B = l->batch; o->cap = l->cap;
    o->PY = (o->Y + 7) / 8; o->PX = (o->X + 7) / 8; o->words = (int)words;
    o->sorted = (l->flags & AQ3D_LEVEL_SORTED) ? 1 : 0;
    o->bits = (unsigned long long *)l->bits; o->base = l->base; o->rank2row = l->row_of_rank; o->coords = l->coords;
    o->count = l->count; o->total = l->total;
    return AQ3D_OK;
}

__device__ __forceinline__ int lv_word(const Lv &L, int b, int z, int py, int px) { return ((b * L.Z + z) * L.PY + py) * L.PX + px; }

// row of the site (b, z, y, x), or -1. Branch-free: the word, its base and the row are loaded whatever the outcome (word 0 /
// rank 0 stand in), so that independent lookups of one thread overlap instead of waiting for each other's bit test.
__device__ __forceinline__ int lv_find(const Lv &L, int b, int z, int y, int x) {
    const bool in = (unsigned)z < (unsigned)L.Z && (unsigned)y < (unsigned)L.Y && (unsigned)x < (unsigned)L.X && (unsigned)b < (unsigned)L.B;
    const int w = in ? lv_word(L, b, z, y >> 3, x >> 3) : 0;
    const unsigned long long bits = L.bits[w];
    const int base = L.base[w];
    const int bit = ((y & 7) << 3) | (x & 7);
    const int rank = base + __popcll(bits & ((1ull << bit) - 1ull));
    const bool hit = in && ((bits >> bit) & 1ull) && rank < L.cap;
    const int row = L.sorted ? L.rank2row[hit ? rank : 0] : rank;      // raster-ordered levels: row == rank (one dependent load less)
    return hit ? row : -1;
}

__device__ __forceinline__ bool lv_test(const Lv &L, int b, int z, int y, int x) {
    if ((unsigned)z >= (unsigned)L.Z || (unsigned)y >= (unsigned)L.Y || (unsigned)x >= (unsigned)L.X) return false;
    return (L.bits[lv_word(L, b, z, y >> 3, x >> 3)] >> (((y & 7) << 3) | (x & 7))) & 1ull;
}

__device__ __forceinline__ void lv_decode(const Lv &L, int cell, int &b, int &z, int &y, int &x) {
    int w = cell >> 6;
    const int bit = cell & 63;
    const int px = w % L.PX; w /= L.PX;
    const int py = w % L.PY; w /= L.PY;
    z = w % L.Z; b = w / L.Z;
    y = py * 8 + (bit >> 3); x = px * 8 + (bit & 7);
}

// ---- block-wide exclusive scan of one int per thread (256 threads); returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int ix_block_scan(int v, int *lds4, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();                    // lds4 may still be read from a previous call
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    int wb = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < IX_BLOCK / 64; ++w) {
        const int s = lds4[w];
        if (w < wave) wb += s;
        tot += s;
    }
    *total = tot;
    return wb + incl - v;
}

// One value per workgroup -> exclusive scan over the workgroups, done by whichever workgroup arrives LAST (no spinning,
// no second launch). Hand-off form of MI355X_MICROARCH.md "Valid forms", row 1: write-through (agent-scope) stores by the
// signalling wave, its s_waitcnt vmcnt(0), an agent-scope ticket add by one lane; the workgroup whose add returns last reads
// with agent-scope loads behind its own barrier. Returns true in the last workgroup only.
// CONTRACT: every store that the last workgroup's scan reads must be issued by WAVE 0 of the signalling workgroup, before
// this call — the s_waitcnt below drains wave 0's stores only. ix_count stores with threads 0 .. IX_CPB-1 (static_assert there),
// ix_dilate with thread 0. A store moved to another wave needs a __syncthreads() + release fence in front of the ticket instead.
#define IX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ bool ix_arrive(int *ticket, int nblocks, int *lds_flag) {
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(ticket, 1, IX_AGENT);
        const int last = (t == nblocks - 1) ? 1 : 0;
        if (last) {      // cdna_hip_programming.md Guideline 16, R1 consumer: ONE agent acquire after the match, its wait, the barrier, then plain loads
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *lds_flag = last;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// last workgroup: chunk_sums[0..n) -> exclusive prefix in place (every thread owns a contiguous segment: one round of independent
// loads, one block scan, one round of stores); publishes total / count, re-arms the ticket
__device__ __forceinline__ void ix_scan_chunks(int *chunk_sums, int n, const Lv &L, int *ticket, int *lds4) {
    const int per = (n + IX_BLOCK - 1) / IX_BLOCK;
    const int i0 = threadIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    int s = 0;
    for (int i = i0; i < i1; i += 8) {          // tiles of 8 independent loads: one memory round trip per tile, not per element
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = i + j < i1 ? chunk_sums[i + j] : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    int tot;
    int run = ix_block_scan(s, lds4, &tot);
    for (int i = i0; i < i1; i += 8) {
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = i + j < i1 ? chunk_sums[i + j] : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (i + j < i1) chunk_sums[i + j] = run;
            run += v[j];
        }
    }
    if (threadIdx.x == 0) {
        L.total[0] = tot;
        L.total[1] = L.B == 1 ? tot : 0;        // per-sample maximum: the total itself for one sample, else ix_offsets / ix_sample_max
        *L.count = tot < L.cap ? tot : L.cap;
        __hip_atomic_store(ticket, 0, IX_AGENT);
    }
}

// ------------------------------------------------------------------ level 0: sites from points / from coordinates
struct MarkParams {
    float min_x, min_y, min_z, vs_x, vs_y, vs_z;
    int nx, ny, nz, stride, col0, batch_col, batch_index;
};

// Clearing the bitmap of a level that is rebuilt every frame (the voxel level). A level's bitmap, coords and count always describe the
// SAME site set (they are written together by one builder), so the words that are set are exactly the words of coords[0 .. count):
// zeroing those is the whole job — 30 000 scattered 8-byte stores instead of an 11.6 MB fill of a grid that is 99.97 % empty.
// (Round 4, scratch/ablate.py: that fill, 8 us on its own, cost 0.25 ms of every 1.04 ms frame interval with four frames in flight;
// no other kernel of the frame is as expensive per microsecond.) A builder that overflowed the row capacity left sites without a
// coords row: then (total > cap) every workgroup falls back to zeroing the whole bitmap.
__global__ __launch_bounds__(IX_BLOCK) void ix_clear_prev(Lv L, int *__restrict__ chunk_flag, int nchunks) {
    for (int i = blockIdx.x * IX_BLOCK + threadIdx.x; i < nchunks; i += gridDim.x * IX_BLOCK) chunk_flag[i] = 0;
    const int total = L.total[0];
    if (total > L.cap) {
        for (size_t w = (size_t)blockIdx.x * IX_BLOCK + threadIdx.x; w < (size_t)L.words; w += (size_t)gridDim.x * IX_BLOCK) L.bits[w] = 0ull;
        return;
    }
    int n = *L.count;
    if (n > L.cap) n = L.cap;
    for (int r = blockIdx.x * IX_BLOCK + threadIdx.x; r < n; r += gridDim.x * IX_BLOCK) {
        const int4 c = reinterpret_cast<const int4 *>(L.coords)[r];
        if ((unsigned)c.x < (unsigned)L.B && (unsigned)c.y < (unsigned)L.Z && (unsigned)c.z < (unsigned)L.Y && (unsigned)c.w < (unsigned)L.X)
            L.bits[lv_word(L, c.x, c.y, c.z >> 3, c.w >> 3)] = 0ull;        // several sites of a word store the same zero: benign
    }
}

__global__ __launch_bounds__(IX_BLOCK) void ix_mark_points(const float *__restrict__ points, int n, MarkParams q, Lv L,
                                                           int *__restrict__ pcell, int *__restrict__ chunk_flag) {
    for (int i = blockIdx.x * IX_BLOCK + threadIdx.x; i < n; i += gridDim.x * IX_BLOCK) {
        const float *p = points + (size_t)i * q.stride;
        const int b = q.batch_col ? (int)p[0] : q.batch_index;
        p += q.col0;
        // IEEE fp32 subtract, divide, floor: the integer cell must be bit-exact (data_processor.py:45-61 -> spconv)
        const float fx = floorf(__fdiv_rn(__fsub_rn(p[0], q.min_x), q.vs_x));
        const float fy = floorf(__fdiv_rn(__fsub_rn(p[1], q.min_y), q.vs_y));
        const float fz = floorf(__fdiv_rn(__fsub_rn(p[2], q.min_z), q.vs_z));
        const bool ok = (fx >= 0.f) && (fx < (float)q.nx) && (fy >= 0.f) && (fy < (float)q.ny) && (fz >= 0.f) &&
                        (fz < (float)q.nz) && b >= 0 && b < L.B;
        int cell = -1;
        if (ok) {
            const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
            if (cz < L.Z && cy < L.Y && cx < L.X) {
                const int w = lv_word(L, b, cz, cy >> 3, cx >> 3), bit = ((cy & 7) << 3) | (cx & 7);
                atomicOr(&L.bits[w], 1ull << bit);
                chunk_flag[w / IX_BLOCK] = 1;             // (many points store the same 1: benign)
                cell = w * 64 + bit;
            }
        }
        pcell[i] = cell;
    }
}

__global__ __launch_bounds__(IX_BLOCK) void ix_mark_coords(const int *__restrict__ coords, int n, Lv L, int *__restrict__ pcell,
                                                           int *__restrict__ chunk_flag) {
    for (int i = blockIdx.x * IX_BLOCK + threadIdx.x; i < n; i += gridDim.x * IX_BLOCK) {
        const int4 c = reinterpret_cast<const int4 *>(coords)[i];
        int cell = -1;
        if (c.x >= 0 && c.x < L.B && (unsigned)c.y < (unsigned)L.Z && (unsigned)c.z < (unsigned)L.Y && (unsigned)c.w < (unsigned)L.X) {
            const int w = lv_word(L, c.x, c.y, c.z >> 3, c.w >> 3), bit = ((c.z & 7) << 3) | (c.w & 7);
            atomicOr(&L.bits[w], 1ull << bit);
            chunk_flag[w / IX_BLOCK] = 1;
            cell = w * 64 + bit;
        }
        pcell[i] = cell;
    }
}

// sites per 256-word chunk + (last workgroup) their exclusive scan. A workgroup counts IX_CPB chunks and takes ONE ticket: the
// arrival counter is a single word (the voxel level of a Waymo grid has 5 664 chunks). Measured: 8 or 32 chunks per workgroup and
// 8- or 16-byte loads all give 21-24 us for that level — the pass is bound by reading the 11.6 MB bitmap the memset and the
// points' atomics have just written, not by the tickets.
#define IX_CPB 8
static_assert(IX_CPB <= 64, "ix_count's chunk sums are stored by threads 0 .. IX_CPB-1, which must all be in wave 0 (ix_arrive's contract)");
__global__ __launch_bounds__(IX_BLOCK) void ix_count(Lv L, IxWs ws, int nchunks, const int *__restrict__ chunk_flag) {
    __shared__ int lds4[IX_BLOCK / 64];
    __shared__ int part[IX_CPB][2];
    __shared__ int flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 16-byte loads: a thread takes two adjacent words, the workgroup two chunks per round (waves 0-1 the first, 2-3 the second)
#pragma unroll
    for (int jj = 0; jj < IX_CPB / 2; ++jj) {
        const int w = (blockIdx.x * IX_CPB + 2 * jj) * IX_BLOCK + 2 * threadIdx.x;
        int pc = 0;
        // level-0 builders: a chunk no point marked is known to be empty (wave-uniform: waves 0-1 / 2-3 share a chunk) — of the voxel
        // level's 5 664 chunks of a Waymo frame some 10 % hold a site
        const int chunk_ = blockIdx.x * IX_CPB + 2 * jj + (wave >> 1);
        if (chunk_flag && chunk_ < nchunks && !chunk_flag[chunk_]) {
        } else if (w + 1 < L.words) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(L.bits + w);
            pc = __popcll(v.x) + __popcll(v.y);
        } else if (w < L.words) pc = __popcll(L.bits[w]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pc += __shfl_xor(pc, off, 64);
        if (lane == 0) part[2 * jj + (wave >> 1)][wave & 1] = pc;
    }
    __syncthreads();
    if (threadIdx.x < IX_CPB && blockIdx.x * IX_CPB + threadIdx.x < nchunks)
        __hip_atomic_store(&ws.chunk_sums[blockIdx.x * IX_CPB + threadIdx.x], part[threadIdx.x][0] + part[threadIdx.x][1], IX_AGENT);
    if (ix_arrive(ws.ticket, gridDim.x, &flag)) ix_scan_chunks(ws.chunk_sums, nchunks, L, ws.ticket, lds4);
}

// ------------------------------------------------------------------ strided / dilating conv: output bitmap from the input bitmap
struct Geo { int kz, ky, kx, sz, sy, sx, pz, py, px; };

__device__ __forceinline__ unsigned ix_even_bits(unsigned v) {     // bits 0, 2, 4, ... 14 -> bits 0..7
    v &= 0x5555u;
    v = (v | (v >> 1)) & 0x3333u;
    v = (v | (v >> 2)) & 0x0f0fu;
    v = (v | (v >> 4)) & 0x00ffu;
    return v;
}

// contribution of input slice iz to the 8x8 output patch (opy, opx): bit (oy, ox) = OR over (ky, kx) of in(oy*sy - py + ky, ox*sx - px + kx)
__device__ __forceinline__ unsigned long long ix_dilate_plane(const Lv &In, int b, int iz, int opy, int opx, const Geo &g) {
    const int y_lo = opy * 8 * g.sy - g.py, x_lo = opx * 8 * g.sx - g.px;
    const int ipy0 = y_lo >> 3, ipx0 = x_lo >> 3;          // arithmetic shifts: floor for -1
    const int shx = x_lo - ipx0 * 8;
    unsigned long long w[3][3];
    unsigned long long any = 0;
#pragma unroll
    for (int jy = 0; jy < 3; ++jy)
#pragma unroll
        for (int jx = 0; jx < 3; ++jx) {
            const int py = ipy0 + jy, px = ipx0 + jx;
            unsigned long long v = 0;
            if (py >= 0 && py < In.PY && px >= 0 && px < In.PX) v = In.bits[lv_word(In, b, iz, py, px)];
            w[jy][jx] = v;
            any |= v;
        }
    if (!any) return 0ull;
    const int nrows = 7 * g.sy + g.ky;                      // <= 17 input rows feed the patch
    unsigned long long out = 0;
#pragma unroll
    for (int t = 0; t < 17; ++t) {
        if (t < nrows) {
            const int iy = y_lo + t;
            const int jy = (iy >> 3) - ipy0, br = (iy & 7) * 8;
            const unsigned long long a0 = jy == 0 ? w[0][0] : (jy == 1 ? w[1][0] : w[2][0]);
            const unsigned long long a1 = jy == 0 ? w[0][1] : (jy == 1 ? w[1][1] : w[2][1]);
            const unsigned long long a2 = jy == 0 ? w[0][2] : (jy == 1 ? w[1][2] : w[2][2]);
            const unsigned r24 = (unsigned)((a0 >> br) & 0xffull) | ((unsigned)((a1 >> br) & 0xffull) << 8) |
                                 ((unsigned)((a2 >> br) & 0xffull) << 16);
            const unsigned E = r24 >> shx;                  // bit c = input column x_lo + c
            const unsigned F = E | (g.kx > 1 ? E >> 1 : 0u) | (g.kx > 2 ? E >> 2 : 0u);
            const unsigned G = g.sx == 1 ? (F & 0xffu) : ix_even_bits(F);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int d = t - ky;
                if (ky < g.ky && d >= 0 && (g.sy == 1 || (d & 1) == 0)) {
                    const int oy = g.sy == 1 ? d : d >> 1;
                    if (oy < 8) out |= (unsigned long long)G << (8 * oy);
                }
            }
        }
    }
    return out;
}

__device__ __forceinline__ unsigned long long ix_valid_mask(const Lv &L, int py, int px) {
    const int rows = L.Y - py * 8 < 8 ? L.Y - py * 8 : 8, cols = L.X - px * 8 < 8 ? L.X - px * 8 : 8;
    const unsigned long long rowm = cols >= 8 ? 0xffull : ((1ull << cols) - 1ull);
    unsigned long long m = 0;
    for (int r = 0; r < rows; ++r) m |= rowm << (8 * r);
    return m;
}

// The same for the common window — 3x3 in (y, x), stride 2, padding 1 — on whole words: every row of a patch is dilated and
// compressed by 64-bit mask / shift operations (8 rows per instruction instead of one), ~5x fewer instructions than the row loop.
__device__ __forceinline__ unsigned long long ix_even_cols(unsigned long long v) {        // bits 0, 2, 4, 6 of every byte -> its low nibble
    v &= 0x5555555555555555ull;
    v = (v | (v >> 1)) & 0x3333333333333333ull;
    v = (v | (v >> 2)) & 0x0f0f0f0f0f0f0f0full;
    return v;
}
__device__ __forceinline__ unsigned long long ix_even_rows(unsigned long long v) {        // bytes 0, 2, 4, 6 -> the low 32 bits
    v &= 0x00ff00ff00ff00ffull;
    v = (v | (v >> 8)) & 0x0000ffff0000ffffull;
    v = (v | (v >> 16)) & 0x00000000ffffffffull;
    return v;
}
__device__ __forceinline__ unsigned long long ix_dilate_plane_s2(const Lv &In, int b, int iz, int opy, int opx) {
    unsigned long long H[3];
    unsigned long long any = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {                        // input patch rows 2 opy - 1, 2 opy, 2 opy + 1
        const int py = 2 * opy - 1 + a;
        unsigned long long A = 0, B = 0, C = 0;         // input patch columns 2 opx - 1, 2 opx, 2 opx + 1
        if (py >= 0 && py < In.PY) {
            const int w = lv_word(In, b, iz, py, 0), px = 2 * opx;
            if (px - 1 >= 0) A = In.bits[w + px - 1];
            if (px < In.PX) B = In.bits[w + px];
            if (px + 1 < In.PX) C = In.bits[w + px + 1];
        }
        any |= A | B | C;
        // bit c of tB = B[c-1] | B[c] | B[c+1] (column -1 = column 7 of A); of tC likewise with column 7 of B
        const unsigned long long tB = B | ((B >> 1) & 0x7f7f7f7f7f7f7f7full) | ((B << 1) & 0xfefefefefefefefeull) | ((A >> 7) & 0x0101010101010101ull);
        const unsigned long long tC = C | ((C >> 1) & 0x7f7f7f7f7f7f7f7full) | ((C << 1) & 0xfefefefefefefefeull) | ((B >> 7) & 0x0101010101010101ull);
        H[a] = ix_even_cols(tB) | (ix_even_cols(tC) << 4);          // output columns 0-3 | 4-7 of the patch row's 8 input rows
    }
    if (!any) return 0ull;
    // rows: output row oy <- input rows 2 oy - 1, 2 oy, 2 oy + 1 (row -1 = row 7 of the patch above)
    const unsigned long long vM = H[1] | (H[1] >> 8) | (H[1] << 8) | (H[0] >> 56);
    const unsigned long long vN = H[2] | (H[2] >> 8) | (H[2] << 8) | (H[1] >> 56);
    return ix_even_rows(vM) | (ix_even_rows(vN) << 32);
}

#define IX_DPB 1        // 256-word chunks per workgroup of ix_dilate (4 per workgroup: 12 -> 25 us, the words of a thread run serially)
__global__ __launch_bounds__(IX_BLOCK) void ix_dilate(Lv In, Lv Out, Geo g, IxWs ws, int nchunks) {
    __shared__ int lds4[IX_BLOCK / 64];
    __shared__ int flag;
    const bool s2 = g.ky == 3 && g.kx == 3 && g.sy == 2 && g.sx == 2 && g.py == 1 && g.px == 1;       // wave-uniform
    const bool flat = g.ky == 1 && g.kx == 1 && g.sy == 1 && g.sx == 1 && g.py == 0 && g.px == 0;     // window in z only
    for (int j = 0; j < IX_DPB; ++j) {
        const int chunk = blockIdx.x * IX_DPB + j;
        if (chunk >= nchunks) break;                     // uniform per workgroup
        const int w = chunk * IX_BLOCK + threadIdx.x;
        int pc = 0;
        if (w < Out.words) {
            int t = w;
            const int px = t % Out.PX; t /= Out.PX;
            const int py = t % Out.PY; t /= Out.PY;
            const int oz = t % Out.Z, b = t / Out.Z;
            unsigned long long bits = 0;
#pragma unroll
            for (int kz = 0; kz < 3; ++kz) {
                const int iz = oz * g.sz - g.pz + kz;
                if (kz < g.kz && iz >= 0 && iz < In.Z)
                    bits |= s2 ? ix_dilate_plane_s2(In, b, iz, py, px) : (flat ? In.bits[lv_word(In, b, iz, py, px)] : ix_dilate_plane(In, b, iz, py, px, g));
            }
            bits &= ix_valid_mask(Out, py, px);
            Out.bits[w] = bits;
            pc = __popcll(bits);
        }
        int tot;
        (void)ix_block_scan(pc, lds4, &tot);
        if (threadIdx.x == 0) __hip_atomic_store(&ws.chunk_sums[chunk], tot, IX_AGENT);
    }
    if (ix_arrive(ws.ticket, gridDim.x, &flag)) ix_scan_chunks(ws.chunk_sums, nchunks, Out, ws.ticket, lds4);
}

// ------------------------------------------------------------------ VoxelNeXt bev_out: union of <= 3 levels projected onto (y, x)
struct MergeSrc {
    Lv L[3];
    const float *feats[3];
    int scale[3];
    int n, C;
};

// one wave per output word, one lane per cell (the 2-D level has few words and every cell asks <= 3 levels x Z slices)
__global__ __launch_bounds__(IX_BLOCK) void ix_merge_bits(MergeSrc S, Lv Out) {
    const int w = blockIdx.x * (IX_BLOCK / 64) + (threadIdx.x >> 6), bit = threadIdx.x & 63;
    if (w >= Out.words) return;
    int t = w;
    const int px = t % Out.PX; t /= Out.PX;
    const int py = t % Out.PY; t /= Out.PY;
    const int b = t / Out.Z;             // Out.Z == 1
    const int y = py * 8 + (bit >> 3), x = px * 8 + (bit & 7);
    bool hit = false;
    if (y < Out.Y && x < Out.X) {
        for (int l = 0; l < S.n; ++l) {
            const int sc = S.scale[l];
            if (y % sc != 0 || x % sc != 0) continue;
            const int iy = y / sc, ix = x / sc;
            if (iy >= S.L[l].Y || ix >= S.L[l].X) continue;
            const int pw = lv_word(S.L[l], b, 0, iy >> 3, ix >> 3), pb = ((iy & 7) << 3) | (ix & 7);
            const int zstride = S.L[l].PY * S.L[l].PX;
            for (int z = 0; z < S.L[l].Z; ++z) hit |= (S.L[l].bits[pw + z * zstride] >> pb) & 1ull;
        }
    }
    const unsigned long long bits = __builtin_amdgcn_ballot_w64(hit);
    if (bit == 0) Out.bits[w] = bits;
}

// thread = (output row, 4 channels): sum of the contributing rows in (level, z) order — no float atomics, bit-reproducible
// (VoxelResBackBone8xVoxelNeXt.bev_out, spconv_backbone_voxelnext.py:149-198: torch.unique + index_add_)
__global__ __launch_bounds__(IX_BLOCK) void ix_merge_sum(MergeSrc S, Lv Out, float *__restrict__ feats_out) {
    int n = *Out.count;
    if (n > Out.cap) n = Out.cap;
    const int c4n = S.C >> 2;
    const long long total = (long long)n * c4n;
    for (long long t = (long long)blockIdx.x * IX_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * IX_BLOCK) {
        const int row = (int)(t / c4n), c4 = (int)(t - (long long)row * c4n);
        const int4 c = reinterpret_cast<const int4 *>(Out.coords)[row];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < S.n; ++l) {
            const int sc = S.scale[l];
            if (c.z % sc != 0 || c.w % sc != 0) continue;
            const int y = c.z / sc, x = c.w / sc;
            for (int z = 0; z < S.L[l].Z; ++z) {
                const int r = lv_find(S.L[l], c.x, z, y, x);
                if (r < 0) continue;
                const float4 v = *reinterpret_cast<const float4 *>(S.feats[l] + (size_t)r * S.C + c4 * 4);
                acc.x = __fadd_rn(acc.x, v.x); acc.y = __fadd_rn(acc.y, v.y);
                acc.z = __fadd_rn(acc.z, v.z); acc.w = __fadd_rn(acc.w, v.w);
            }
        }
        *reinterpret_cast<float4 *>(feats_out + (size_t)row * S.C + c4 * 4) = acc;
    }
}

// ------------------------------------------------------------------ ranks -> class-sorted rows
// base[] of every word, cell of every rank, first rank of every sample
__global__ __launch_bounds__(IX_BLOCK) void ix_rank(Lv L, IxWs ws, int nchunks, const int *__restrict__ chunk_flag) {
    __shared__ int lds4[IX_BLOCK / 64];
    // a workgroup walks several 256-word chunks (round 4): the voxel level of a Waymo grid has 5 664 chunks of which ~10 % hold a site,
    // and under four frames in flight a kernel pays for every workgroup it asks the dispatcher for, not for the work inside
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int w = chunk * IX_BLOCK + threadIdx.x;
        if (chunk_flag && !chunk_flag[chunk]) {      // (uniform) an empty chunk: no site to number, and base[] of an empty word is never used
            const int per_sample = L.words / L.B;    // ... except that a sample may START here
            if (w < L.words)
                for (int b = 0; b < L.B; ++b)
                    if (w == b * per_sample) ws.sstart[b] = ws.chunk_sums[chunk];
            continue;
        }
        unsigned long long v = w < L.words ? L.bits[w] : 0ull;
        int tot;
        int r = ws.chunk_sums[chunk] + ix_block_scan(__popcll(v), lds4, &tot);
        if (w >= L.words) continue;
        L.base[w] = r;
        const int per_sample = L.words / L.B;
        for (int b = 0; b < L.B; ++b)           // (no per-thread division: nearly every word of a level is empty and its thread ends here)
            if (w == b * per_sample) ws.sstart[b] = r;
        if (!v) continue;
        if (L.sorted) {
            while (v) {
                const int bit = __builtin_ctzll(v);
                if (r < L.cap) ws.cell_of_rank[r] = w * 64 + bit;
                ++r;
                v &= v - 1ull;
            }
            continue;
        }
        // raster-ordered level: row = rank, the level is complete after this kernel
        int t = w;
        const int px = t % L.PX; t /= L.PX;
        const int py = t % L.PY; t /= L.PY;
        const int z = t % L.Z, b = t / L.Z;
        while (v) {
            const int bit = __builtin_ctzll(v);
            if (r < L.cap) {
                L.rank2row[r] = r;
                reinterpret_cast<int4 *>(L.coords)[r] = make_int4(b, z, py * 8 + (bit >> 3), px * 8 + (bit & 7));
                ws.head[r] = -1;
            }
            ++r;
            v &= v - 1ull;
        }
    }
}

// batch > 1, raster-ordered level: the largest per-sample number of sites (MAX_NUMBER_OF_VOXELS is a per-sample limit)
__global__ void ix_sample_max(Lv L, IxWs ws) {
    if ((int)threadIdx.x < L.B) {
        const int s0 = ws.sstart[threadIdx.x], s1 = (int)threadIdx.x + 1 < L.B ? ws.sstart[threadIdx.x + 1] : L.total[0];
        atomicMax(&L.total[1], s1 - s0);
    }
}

// Class of a site: which planes of its 3x3x3 neighbourhood hold any site — 3 bits for kz = 0, 1, 2, 2 bits for ky = 0, 2, 2 bits
// for kx = 0, 2. Rows of one class can only have neighbours at |kz set| * (1 + |ky set|) * (1 + |kx set|) of the 27 offsets.
// Classes are laid out heaviest first so that the tiles with the most work are dispatched first.
constexpr int ix_pop3(int v) { return (v & 1) + ((v >> 1) & 1) + ((v >> 2) & 1); }
constexpr int ix_class_weight(int key) { return ix_pop3(key & 7) * (1 + ix_pop3((key >> 3) & 3)) * (1 + ix_pop3((key >> 5) & 3)); }
struct IxClassOrder { unsigned char pos[128]; };
constexpr IxClassOrder ix_make_class_order() {          // position of class `key`: descending weight, ties by key
    IxClassOrder t{};
    for (int k = 0; k < 128; ++k) {
        int p = 0;
        for (int j = 0; j < 128; ++j) {
            const int wj = ix_class_weight(j), wk = ix_class_weight(k);
            p += (wj > wk || (wj == wk && j < k)) ? 1 : 0;
        }
        t.pos[k] = (unsigned char)p;
    }
    return t;
}
__constant__ IxClassOrder IX_ORDER = ix_make_class_order();

__global__ __launch_bounds__(IX_BLOCK) void ix_class(Lv L, IxWs ws) {
    __shared__ int hist[128];
    int n = L.total[0];
    if (n > L.cap) n = L.cap;
    if ((int)blockIdx.x * IX_BLOCK >= n) return;
    if (threadIdx.x < 128) hist[threadIdx.x] = 0;
    __syncthreads();
    const int r = blockIdx.x * IX_BLOCK + threadIdx.x;
    if (r < n) {
        int b, z, y, x;
        lv_decode(L, ws.cell_of_rank[r], b, z, y, x);
        int key = 0;
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx)
                    if (lv_test(L, b, z + dz, y + dy, x + dx)) {
                        key |= 1 << (dz + 1);
                        if (dy != 0) key |= dy < 0 ? 8 : 16;
                        if (dx != 0) key |= dx < 0 ? 32 : 64;
                    }
        ws.key_of_rank[r] = (unsigned char)key;
        atomicAdd(&hist[key], 1);
    }
    __syncthreads();
    if (threadIdx.x < 128) ws.hist[blockIdx.x * 128 + threadIdx.x] = hist[threadIdx.x];
}

// workgroup c: exclusive prefix of class c's counts over the 256-rank blocks (in place) and the class total
__global__ __launch_bounds__(IX_BLOCK) void ix_offsets(Lv L, IxWs ws) {
    __shared__ int lds4[IX_BLOCK / 64];
    int n = L.total[0];
    if (n > L.cap) n = L.cap;
    const int nblk = (n + IX_BLOCK - 1) / IX_BLOCK, c = blockIdx.x;
    int carry = 0;
    for (int base = 0; base < nblk; base += IX_BLOCK) {
        const int blk = base + threadIdx.x;
        const int v = blk < nblk ? ws.hist[blk * 128 + c] : 0;
        int tot;
        const int ex = ix_block_scan(v, lds4, &tot);
        if (blk < nblk) ws.hist[blk * 128 + c] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) ws.cls_tot[c] = carry;
    // per-sample totals (max_voxels is a per-sample limit, data_processor.py:45-61)
    if (c == 0 && L.B > 1 && threadIdx.x < L.B) {
        const int s0 = ws.sstart[threadIdx.x], s1 = threadIdx.x + 1 < L.B ? ws.sstart[threadIdx.x + 1] : L.total[0];
        atomicMax(&L.total[1], s1 - s0);
    }
}

// rank -> row (stable within a class), coords of every row, empty point list per row
__global__ __launch_bounds__(IX_BLOCK) void ix_scatter(Lv L, IxWs ws) {
    __shared__ int wcnt[IX_BLOCK / 64][128];
    __shared__ int cls_start[128];
    __shared__ int cls_n[128];
    int n = L.total[0];
    if (n > L.cap) n = L.cap;
    const int r = blockIdx.x * IX_BLOCK + threadIdx.x;
    if (blockIdx.x * IX_BLOCK >= n) return;
    for (int i = threadIdx.x; i < (IX_BLOCK / 64) * 128; i += IX_BLOCK) (&wcnt[0][0])[i] = 0;
    // first row of every class: exclusive scan of the class totals in class order (IX_ORDER), every workgroup for itself
    if (threadIdx.x < 128) cls_n[IX_ORDER.pos[threadIdx.x]] = ws.cls_tot[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 128) {
        const int v = cls_n[threadIdx.x];
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if ((threadIdx.x & 63) >= off) incl += t;
        }
        cls_start[threadIdx.x] = incl - v;                   // by position, within the wave's 64 positions
        if (threadIdx.x == 63) cls_n[0] = incl;              // total of positions 0..63 (cls_n[0] has been read by its lane)
    }
    __syncthreads();
    const int half0 = cls_n[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool valid = r < n;
    const int key = valid ? (int)ws.key_of_rank[r] : -1;
    int lrank = 0;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int k0 = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(valid && key == k0);
        if (key == k0) lrank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == leader) wcnt[wave][k0] = __popcll(m);
        todo &= ~m;
    }
    __syncthreads();
    if (!valid) return;
    const int pos = IX_ORDER.pos[key];
    int row = cls_start[pos] + (pos >= 64 ? half0 : 0) + ws.hist[blockIdx.x * 128 + key] + lrank;
    for (int w = 0; w < wave; ++w) row += wcnt[w][key];
    L.rank2row[r] = row;
    int b, z, y, x;
    lv_decode(L, ws.cell_of_rank[r], b, z, y, x);
    reinterpret_cast<int4 *>(L.coords)[row] = make_int4(b, z, y, x);
    ws.head[row] = -1;
}

// ------------------------------------------------------------------ gather tables
struct NbrJob {
    Lv In;
    Geo g;
    int *nbr;
    int kvol;
};
struct NbrJobs {
    NbrJob j[4];
    int n, ktotal;
};

__device__ __forceinline__ int ix_div_small(int k, int d) {      // k < 256, d in 1..3
    return d == 1 ? k : (d == 2 ? k >> 1 : (k * 171) >> 9);
}

// A wave serves 64 >> lg consecutive rows, 1 << lg lanes each (the padded width of the rows' ktotal table entries): consecutive
// lanes share a site, the table rows are written coalesced, and no index needs a run-time division.
__global__ __launch_bounds__(IX_BLOCK) void ix_neighbors(Lv Out, NbrJobs J, int lg) {
    // the row count is only needed by the stores: rows up to the capacity are looked up whatever it says (coordinates beyond the live
    // rows are stale or zero: lv_find checks every component), so the count's round trip overlaps the coordinate loads instead of
    // preceding them
    int n = *Out.count;
    if (n > Out.cap) n = Out.cap;
    const int lane = threadIdx.x & 63;
    const int rpw = 64 >> lg, sub = lane >> lg, kk = lane & ((1 << lg) - 1);
    const int nwaves = gridDim.x * (IX_BLOCK / 64);
    if (kk >= J.ktotal) return;
    int k = kk, ji = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (ji == q && q + 1 < J.n && k >= J.j[q].kvol) { k -= J.j[q].kvol; ji = q + 1; }
    // the job's fields one by one through selects (a dynamic index into the kernel argument, or a selected struct, goes through memory)
#define IX_SEL(f) (ji == 0 ? J.j[0].f : (ji == 1 ? J.j[1].f : (ji == 2 ? J.j[2].f : J.j[3].f)))
    Lv In;
    In.Z = IX_SEL(In.Z); In.Y = IX_SEL(In.Y); In.X = IX_SEL(In.X); In.B = IX_SEL(In.B); In.cap = IX_SEL(In.cap);
    In.PY = IX_SEL(In.PY); In.PX = IX_SEL(In.PX); In.words = 0; In.sorted = IX_SEL(In.sorted);
    In.bits = IX_SEL(In.bits); In.base = IX_SEL(In.base); In.rank2row = IX_SEL(In.rank2row);
    In.coords = nullptr; In.count = nullptr; In.total = nullptr;
    Geo g;
    g.kz = IX_SEL(g.kz); g.ky = IX_SEL(g.ky); g.kx = IX_SEL(g.kx); g.sz = IX_SEL(g.sz); g.sy = IX_SEL(g.sy); g.sx = IX_SEL(g.sx);
    g.pz = IX_SEL(g.pz); g.py = IX_SEL(g.py); g.px = IX_SEL(g.px);
    int *nbr = IX_SEL(nbr);
    const int kvol = IX_SEL(kvol);
#undef IX_SEL
    const int kyx = g.ky * g.kx;
    const int kz = kyx == 9 ? (k * 57) >> 9 : (kyx == 1 ? k : k / kyx);          // k < 27: k / 9 = (k * 57) >> 9
    const int rem = k - kz * kyx;
    const int ky = ix_div_small(rem, g.kx), kx = rem - ky * g.kx;
    const int oz = kz - g.pz, oy = ky - g.py, ox = kx - g.px;
    // four rows per lane and round: their lookups are independent chains (coords -> word + base -> row) that overlap
    constexpr int U = 4;
    for (int wv = (blockIdx.x * (IX_BLOCK / 64) + (threadIdx.x >> 6)) * U; wv * rpw < Out.cap; wv += nwaves * U) {
        int m[U];
        int4 c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            m[u] = (wv + u) * rpw + sub;
            c[u] = reinterpret_cast<const int4 *>(Out.coords)[m[u] < Out.cap ? m[u] : 0];
        }
        if (wv * rpw >= n) break;
        int r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = lv_find(In, c[u].x, c[u].y * g.sz + oz, c[u].z * g.sy + oy, c[u].w * g.sx + ox);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (m[u] < n) nbr[(size_t)m[u] * kvol + k] = r[u];
    }
}

// ------------------------------------------------------------------ voxel features (hard voxelization + MeanVFE) in row order
__global__ __launch_bounds__(IX_BLOCK) void ix_link(int n, Lv L, IxWs ws) {
    for (int i = blockIdx.x * IX_BLOCK + threadIdx.x; i < n; i += gridDim.x * IX_BLOCK) {
        const int cell = ws.pcell[i];
        if (cell < 0) continue;
        const int w = cell >> 6, bit = cell & 63;
        const int rank = L.base[w] + __popcll(L.bits[w] & ((1ull << bit) - 1ull));
        if (rank >= L.cap) continue;
        const int row = L.sorted ? L.rank2row[rank] : rank;
        ws.next[i] = atomicExch(&ws.head[row], i);
    }
}

#define IX_PMAX 16
// one thread per row: the P smallest point indices of its list (static compare-swap ladder, registers only), summed in
// point order and divided by their number — bit-exact with "first P points in point order" + MeanVFE (mean_vfe.py:25-29)
__global__ __launch_bounds__(IX_BLOCK) void ix_voxel_features(const float *__restrict__ points, MarkParams q, int C, int P, Lv L,
                                                              IxWs ws, float *__restrict__ feats_out, int *__restrict__ npts_out) {
    int n = *L.count;
    if (n > L.cap) n = L.cap;
    for (int row = blockIdx.x * IX_BLOCK + threadIdx.x; row < n; row += gridDim.x * IX_BLOCK) {
        int idx[IX_PMAX];
#pragma unroll
        for (int j = 0; j < IX_PMAX; ++j) idx[j] = 0x7fffffff;
        int total = 0;
        for (int p = ws.head[row]; p >= 0; p = ws.next[p]) {
            int cand = p;
            ++total;
#pragma unroll
            for (int j = 0; j < IX_PMAX; ++j) {
                const int lo = min(idx[j], cand), hi = max(idx[j], cand);
                idx[j] = lo;
                cand = hi;
            }
        }
        const int kept = total < P ? total : P;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
        for (int j = 0; j < IX_PMAX; ++j) {
            if (j < kept) {
                const float *p = points + (size_t)idx[j] * q.stride + q.col0;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < C) acc[c] = __fadd_rn(acc[c], p[c]);
            }
        }
        const float denom = (float)(kept < 1 ? 1 : kept);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) feats_out[(size_t)row * C + c] = __fdiv_rn(acc[c], denom);
        if (npts_out) npts_out[row] = kept;
    }
}

__global__ __launch_bounds__(IX_BLOCK) void ix_rows_of_cells(int n, Lv L, IxWs ws, int *__restrict__ row_of_input) {
    for (int i = blockIdx.x * IX_BLOCK + threadIdx.x; i < n; i += gridDim.x * IX_BLOCK) {
        const int cell = ws.pcell[i];
        int row = -1;
        if (cell >= 0) {
            const int w = cell >> 6, bit = cell & 63;
            const int rank = L.base[w] + __popcll(L.bits[w] & ((1ull << bit) - 1ull));
            if (rank < L.cap) row = L.rank2row[rank];
        }
        row_of_input[i] = row;
    }
}

// ------------------------------------------------------------------ host side
static int ix_words_grid(const Lv &L) { return aq3d_div_up(L.words, IX_BLOCK); }

// rank / class / scatter of a level whose bitmap and chunk sums are complete
static void ix_number_rows(const Lv &L, const IxWs &ws, hipStream_t stream, const int *chunk_flag = nullptr) {
    const int rank_chunks = ix_words_grid(L);
    hipLaunchKernelGGL(ix_rank, dim3(rank_chunks < 512 ? rank_chunks : 512), dim3(IX_BLOCK), 0, stream, L, ws, rank_chunks, chunk_flag);
    if (!L.sorted) {
        if (L.B > 1) hipLaunchKernelGGL(ix_sample_max, dim3(1), dim3(64), 0, stream, L, ws);
        return;
    }
    const int rblocks = aq3d_div_up(L.cap, IX_BLOCK);
    hipLaunchKernelGGL(ix_class, dim3(rblocks), dim3(IX_BLOCK), 0, stream, L, ws);
    hipLaunchKernelGGL(ix_offsets, dim3(128), dim3(IX_BLOCK), 0, stream, L, ws);
    hipLaunchKernelGGL(ix_scatter, dim3(rblocks), dim3(IX_BLOCK), 0, stream, L, ws);
}

static int ix_mark_params(const aq3d_voxel_cfg *cfg, int batch_index, MarkParams *q) {
    AQ3D_REQUIRE(cfg->num_features >= 3 && cfg->num_features <= 8, "aq3d_level_from_points: num_features=%d not in 3..8", cfg->num_features);
    AQ3D_REQUIRE(cfg->max_points_per_voxel >= 1 && cfg->max_points_per_voxel <= IX_PMAX,
                 "aq3d_level_from_points: max_points_per_voxel=%d not in 1..%d", cfg->max_points_per_voxel, IX_PMAX);
    AQ3D_REQUIRE(cfg->point_stride >= cfg->num_features + (cfg->batch_col ? 1 : 0), "aq3d_level_from_points: point_stride too small");
    q->min_x = cfg->range_min[0]; q->min_y = cfg->range_min[1]; q->min_z = cfg->range_min[2];
    q->vs_x = cfg->voxel_size[0]; q->vs_y = cfg->voxel_size[1]; q->vs_z = cfg->voxel_size[2];
    q->nx = cfg->grid[0]; q->ny = cfg->grid[1]; q->nz = cfg->grid[2];
    q->stride = cfg->point_stride; q->col0 = cfg->batch_col ? 1 : 0; q->batch_col = cfg->batch_col ? 1 : 0; q->batch_index = batch_index;
    return AQ3D_OK;
}

extern "C" int aq3d_level_from_points(const float *points, int32_t n_points, const aq3d_voxel_cfg *cfg, int32_t batch_index,
                                      const aq3d_level *level, float *feats_out, int32_t *npts_out, void *workspace,
                                      size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(cfg && feats_out && n_points >= 0 && (points || n_points == 0), "aq3d_level_from_points: null argument");
    Lv L;
    int rc = lv_of(level, &L, "aq3d_level_from_points");
    if (rc != AQ3D_OK) return rc;
    MarkParams q;
    rc = ix_mark_params(cfg, batch_index, &q);
    if (rc != AQ3D_OK) return rc;
    AQ3D_REQUIRE(cfg->grid[2] <= L.Z && cfg->grid[1] <= L.Y && cfg->grid[0] <= L.X, "aq3d_level_from_points: voxel grid exceeds the level's shape");
    IxWs ws;
    if (!ix_carve(workspace, workspace_bytes, L.cap, L.words, n_points, &ws)) {
        aq3d_set_error("aq3d_level_from_points: workspace %zu < %zu bytes", workspace_bytes, aq3d_index_workspace_bytes(L.cap, L.words, n_points));
        return AQ3D_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(ix_clear_prev, dim3(128), dim3(IX_BLOCK), 0, stream, L, ws.chunk_flag, ix_words_grid(L));       // precondition (aq3d.h): bits / coords / count / total consistent, e.g. all zero
    if (n_points > 0)
        hipLaunchKernelGGL(ix_mark_points, dim3(aq3d_grid(n_points, IX_BLOCK)), dim3(IX_BLOCK), 0, stream, points, n_points, q, L, ws.pcell, ws.chunk_flag);
    hipLaunchKernelGGL(ix_count, dim3(aq3d_div_up(ix_words_grid(L), IX_CPB)), dim3(IX_BLOCK), 0, stream, L, ws, ix_words_grid(L), (const int *)ws.chunk_flag);
    ix_number_rows(L, ws, stream, ws.chunk_flag);
    if (n_points > 0)
        hipLaunchKernelGGL(ix_link, dim3(aq3d_grid(n_points, IX_BLOCK)), dim3(IX_BLOCK), 0, stream, n_points, L, ws);
    hipLaunchKernelGGL(ix_voxel_features, dim3(aq3d_grid(L.cap, IX_BLOCK)), dim3(IX_BLOCK), 0, stream, points, q, cfg->num_features,
                       cfg->max_points_per_voxel, L, ws, feats_out, npts_out);
    AQ3D_CHECK_LAUNCH("aq3d_level_from_points");
    return AQ3D_OK;
}

extern "C" int aq3d_level_from_coords(const int32_t *coords_in, int32_t n, const aq3d_level *level, int32_t *row_of_input,
                                      void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(n >= 0 && (coords_in || n == 0) && (row_of_input || n == 0), "aq3d_level_from_coords: null argument");
    Lv L;
    int rc = lv_of(level, &L, "aq3d_level_from_coords");
    if (rc != AQ3D_OK) return rc;
    IxWs ws;
    if (!ix_carve(workspace, workspace_bytes, L.cap, L.words, n, &ws)) {
        aq3d_set_error("aq3d_level_from_coords: workspace %zu < %zu bytes", workspace_bytes, aq3d_index_workspace_bytes(L.cap, L.words, n));
        return AQ3D_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(ix_clear_prev, dim3(128), dim3(IX_BLOCK), 0, stream, L, ws.chunk_flag, ix_words_grid(L));       // precondition (aq3d.h): bits / coords / count / total consistent, e.g. all zero
    if (n > 0) hipLaunchKernelGGL(ix_mark_coords, dim3(aq3d_grid(n, IX_BLOCK)), dim3(IX_BLOCK), 0, stream, coords_in, n, L, ws.pcell, ws.chunk_flag);
    hipLaunchKernelGGL(ix_count, dim3(aq3d_div_up(ix_words_grid(L), IX_CPB)), dim3(IX_BLOCK), 0, stream, L, ws, ix_words_grid(L), (const int *)ws.chunk_flag);
    ix_number_rows(L, ws, stream, ws.chunk_flag);
    if (n > 0) hipLaunchKernelGGL(ix_rows_of_cells, dim3(aq3d_grid(n, IX_BLOCK)), dim3(IX_BLOCK), 0, stream, n, L, ws, row_of_input);
    AQ3D_CHECK_LAUNCH("aq3d_level_from_coords");
    return AQ3D_OK;
}

static int ix_geo(const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3], Geo *g, const char *who) {
    g->kz = ksize[0]; g->ky = ksize[1]; g->kx = ksize[2];
    g->sz = stride[0]; g->sy = stride[1]; g->sx = stride[2];
    g->pz = pad[0]; g->py = pad[1]; g->px = pad[2];
    AQ3D_REQUIRE(g->kz >= 1 && g->kz <= 3 && g->ky >= 1 && g->ky <= 3 && g->kx >= 1 && g->kx <= 3, "%s: kernel size must be 1..3 per axis", who);
    AQ3D_REQUIRE(g->sz >= 1 && g->sz <= 2 && g->sy >= 1 && g->sy <= 2 && g->sx >= 1 && g->sx <= 2, "%s: stride must be 1 or 2 per axis", who);
    AQ3D_REQUIRE(g->pz >= 0 && g->pz <= 1 && g->py >= 0 && g->py <= 1 && g->px >= 0 && g->px <= 1, "%s: padding must be 0 or 1 per axis", who);
    return AQ3D_OK;
}

extern "C" int aq3d_level_downsample(const aq3d_level *in, const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3],
                                     const aq3d_level *out, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Lv In, Out;
    int rc = lv_of(in, &In, "aq3d_level_downsample(in)");
    if (rc != AQ3D_OK) return rc;
    rc = lv_of(out, &Out, "aq3d_level_downsample(out)");
    if (rc != AQ3D_OK) return rc;
    Geo g;
    rc = ix_geo(ksize, stride, pad, &g, "aq3d_level_downsample");
    if (rc != AQ3D_OK) return rc;
    AQ3D_REQUIRE(Out.B == In.B && Out.Z == (In.Z + 2 * g.pz - g.kz) / g.sz + 1 && Out.Y == (In.Y + 2 * g.py - g.ky) / g.sy + 1 &&
                     Out.X == (In.X + 2 * g.px - g.kx) / g.sx + 1,
                 "aq3d_level_downsample: output shape (%d,%d,%d) is not (in + 2p - k) / s + 1", Out.Z, Out.Y, Out.X);
    IxWs ws;
    if (!ix_carve(workspace, workspace_bytes, Out.cap, Out.words, 0, &ws)) {
        aq3d_set_error("aq3d_level_downsample: workspace %zu < %zu bytes", workspace_bytes, aq3d_index_workspace_bytes(Out.cap, Out.words, 0));
        return AQ3D_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(ix_dilate, dim3(aq3d_div_up(ix_words_grid(Out), IX_DPB)), dim3(IX_BLOCK), 0, stream, In, Out, g, ws, ix_words_grid(Out));
    ix_number_rows(Out, ws, stream);
    AQ3D_CHECK_LAUNCH("aq3d_level_downsample");
    return AQ3D_OK;
}

static int ix_merge_src(const aq3d_level *const *levels, const float *const *feats, const int32_t *scales, int32_t n, int32_t C,
                        const aq3d_level *out, MergeSrc *S, Lv *Out) {
    AQ3D_REQUIRE(levels && scales && n >= 1 && n <= 3, "aq3d_level_merge: 1..3 source levels");
    int rc = lv_of(out, Out, "aq3d_level_merge(out)");
    if (rc != AQ3D_OK) return rc;
    AQ3D_REQUIRE(Out->Z == 1, "aq3d_level_merge: the output level is 2-D (shape[0] == 1)");
    for (int l = 0; l < 3; ++l) {
        const int k = l < n ? l : 0;
        rc = lv_of(levels[k], &S->L[l], "aq3d_level_merge(src)");
        if (rc != AQ3D_OK) return rc;
        AQ3D_REQUIRE(scales[k] >= 1 && S->L[l].B == Out->B, "aq3d_level_merge: scale >= 1, equal batch");
        S->scale[l] = scales[k];
        S->feats[l] = feats ? feats[k] : nullptr;
    }
    S->n = n; S->C = C;
    return AQ3D_OK;
}

extern "C" int aq3d_level_merge(const aq3d_level *const *levels, const int32_t *scales, int32_t n, const aq3d_level *out,
                                void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MergeSrc S;
    Lv Out;
    int rc = ix_merge_src(levels, nullptr, scales, n, 4, out, &S, &Out);
    if (rc != AQ3D_OK) return rc;
    IxWs ws;
    if (!ix_carve(workspace, workspace_bytes, Out.cap, Out.words, 0, &ws)) {
        aq3d_set_error("aq3d_level_merge: workspace too small");
        return AQ3D_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(ix_merge_bits, dim3(aq3d_div_up(Out.words, IX_BLOCK / 64)), dim3(IX_BLOCK), 0, stream, S, Out);
    hipLaunchKernelGGL(ix_count, dim3(aq3d_div_up(ix_words_grid(Out), IX_CPB)), dim3(IX_BLOCK), 0, stream, Out, ws, ix_words_grid(Out), (const int *)nullptr);
    ix_number_rows(Out, ws, stream);
    AQ3D_CHECK_LAUNCH("aq3d_level_merge");
    return AQ3D_OK;
}

extern "C" int aq3d_level_merge_sum(const aq3d_level *const *levels, const float *const *feats, const int32_t *scales, int32_t n,
                                    int32_t C, const aq3d_level *out, float *feats_out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(feats && feats_out && C >= 4 && (C & 3) == 0, "aq3d_level_merge_sum: feats / C");
    MergeSrc S;
    Lv Out;
    int rc = ix_merge_src(levels, feats, scales, n, C, out, &S, &Out);
    if (rc != AQ3D_OK) return rc;
    for (int l = 0; l < n; ++l) AQ3D_REQUIRE(feats[l], "aq3d_level_merge_sum: null feature pointer");
    hipLaunchKernelGGL(ix_merge_sum, dim3(aq3d_grid((int64_t)Out.cap * (C / 4), IX_BLOCK, 4096)), dim3(IX_BLOCK), 0, stream, S, Out, feats_out);
    AQ3D_CHECK_LAUNCH("aq3d_level_merge_sum");
    return AQ3D_OK;
}

extern "C" int aq3d_level_neighbors(const aq3d_level *out, const aq3d_nbr_job *jobs, int32_t n_jobs, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= 4, "aq3d_level_neighbors: 1..4 tables per launch");
    Lv Out;
    int rc = lv_of(out, &Out, "aq3d_level_neighbors(out)");
    if (rc != AQ3D_OK) return rc;
    NbrJobs J;
    J.n = n_jobs; J.ktotal = 0;
    for (int i = 0; i < 4; ++i) {
        const aq3d_nbr_job &s = jobs[i < n_jobs ? i : 0];
        AQ3D_REQUIRE(s.nbr && s.in, "aq3d_level_neighbors: null job field");
        rc = lv_of(s.in, &J.j[i].In, "aq3d_level_neighbors(in)");
        if (rc != AQ3D_OK) return rc;
        rc = ix_geo(s.ksize, s.stride, s.pad, &J.j[i].g, "aq3d_level_neighbors");
        if (rc != AQ3D_OK) return rc;
        J.j[i].nbr = s.nbr;
        J.j[i].kvol = s.ksize[0] * s.ksize[1] * s.ksize[2];
        if (i < n_jobs) J.ktotal += J.j[i].kvol;
    }
    AQ3D_REQUIRE(J.ktotal <= 64, "aq3d_level_neighbors: the tables of one launch hold %d > 64 entries per row", J.ktotal);
    int lg = 0;
    while ((1 << lg) < J.ktotal) ++lg;
    const int rows_per_block = (IX_BLOCK / 64) * (64 >> lg) * 4 * 4;     // four rounds of four rows per wave (the job / tap set-up of a wave costs as much as a round; 1 / 4 / 8 rounds: 15.4 / 12.8 / 16.7 us)
    hipLaunchKernelGGL(ix_neighbors, dim3(aq3d_grid(aq3d_div_up(Out.cap, rows_per_block), 1, 16384)), dim3(IX_BLOCK), 0, stream, Out, J, lg);
    AQ3D_CHECK_LAUNCH("aq3d_level_neighbors");
    return AQ3D_OK;
}
