// Wide-tile sparse conv for the Cin >= 32 layers: 128 output rows per workgroup, activations gathered straight into the
// MFMA A registers, only the weights go through LDS.
//
// Why: with 32-row tiles (spconv_split.hip) every workgroup re-streams the layer's weights (27 x Cout x Cin fp16 = up to
// 884 KB) from L2 and stages activations AND weights through LDS behind two barriers per kernel offset; rocprof shows
// ~1.2 GB of L2 requests per launch for an 18 MB activation tensor, i.e. the kernel is bound by L2 -> LDS staging, two
// thirds of it weights. Here a workgroup (4 waves) owns 128 rows x all Cout columns:
//   * weights: one [Cout x 64-channel] half-slice per unit of work is staged (global -> registers -> LDS, double
//     buffered) and shared by the 4 waves — 4x less weight traffic per output row than the 32-row kernel;
//   * activations: the A operand of v_mfma_f32_16x16x32_f16 is, per lane, 8 consecutive K values of ONE output row, i.e.
//     16 contiguous bytes of ONE gathered feature row, so each lane fetches its own fragments with bounded 16-byte buffer
//     loads (a missing neighbour is an out-of-range offset -> zeros, no memory access, no branch). No LDS, no barrier on
//     the activation path; the loads of unit i+1 fly under the MFMAs of unit i;
//   * each wave computes 2 row tiles x all column tiles, so one weight fragment read from LDS feeds 4 MFMAs
//     (2 row tiles x hi/lo), which keeps the LDS read rate at half the MFMA issue rate.
// Numerics: activations are the (hi, lo) fp16 pair of aq3d_quantize_rows (v = hi + lo/2048), weights are integer codes
// held in fp16. hi*w and (lo)*(w/2048) are exact fp32 products (w/2048 is an exact fp16), so a SINGLE fp32 accumulator
// takes both; the weight fragment is rescaled in registers (4 packed fp16 multiplies per fragment).
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
#define WD_OOB 0xfffffff0u

struct SpconvWideArgs {
    const _Float16 *feats;    // [M_in, 2*CIN]  row = hi(CIN) | lo(CIN)
    const _Float16 *weight;   // [kvol][COUT][CIN] integer codes as fp16
    const float *w_dequant, *bias, *post_scale, *post_shift, *residual;
    const int *nbr;           // [cap_out][kvol]
    const int *count_out;
    float *feats_out, *absmax_out;
    int cap_out, kvol, relu;
    unsigned feats_bytes;
    long long *stamps;        // STAMP builds (AQ3D_WIDE_DEBUG=1): cycle sums over (workgroup, wave 0, unit)
    float *partial;           // split mode scratch: [tiles][2][128 * COUT] accumulator images (NULL: never split)
    int *flags;               // split mode: [tiles] arrival counters, zero between launches (self-resetting)
    int split_below;          // split the kernel offsets of a tile over two workgroups while tiles < split_below
};

template <int CIN, int COUT, bool STAMP>
__global__ __launch_bounds__(256) void spconv_wide_kernel(SpconvWideArgs a) {
    constexpr int RT = 2;                               // 16-row tiles per wave
    constexpr int TM = 4 * RT * 16;                     // rows per workgroup
    constexpr int NT = COUT / 16;
    constexpr int UC = CIN < 64 ? CIN : 64;             // input channels per unit of work
    constexpr int NU = CIN / UC;                        // units per kernel offset
    constexpr int KC = UC / 32;                         // MFMA K chunks per unit (per hi / lo half)
    // LDS weight rows: UC = 64 -> unpadded 128-byte rows with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7, which is
    // conflict-free for ds_read_b128's 16-lane groups ({0-3,12-15,20-27}, ...: 16 distinct rows, chunks g / g+1) and for the
    // 8-lane contiguous groups of ds_write_b128; UC = 32 keeps a 16-byte pad (2-way on the reads of one minor layer).
    constexpr bool SWZ = UC == 64;
    constexpr int LDW = SWZ ? UC : UC + 8;              // halves per LDS weight row
    constexpr int WCH = COUT * UC / 8;                  // 16-byte chunks of one weight half-slice
    constexpr int NWL = (WCH + 255) / 256;              // weight chunks per thread
    static_assert(CIN % 32 == 0 && UC % 32 == 0 && COUT % 16 == 0, "channel granularity");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16 *Wb = reinterpret_cast<_Float16 *>(smem_raw);              // [2][COUT][LDW]
    int *nbr_s = reinterpret_cast<int *>(Wb + 2 * COUT * LDW);          // [TM][kvol]
    float *red = reinterpret_cast<float *>(nbr_s + TM * 27);            // [4][COUT]
    int *mask_s = reinterpret_cast<int *>(red + 4 * COUT);              // [1]: kernel offsets with work in this tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int kvol = a.kvol;
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    // Tile quantisation: with T tiles on 256 CUs the kernel lasts ceil(T / 256) tile times (299 tiles: 2 instead of 1.17).
    // While T is small, TWO workgroups share a tile, each taking half of its live kernel offsets; the one that finishes
    // second adds the other's accumulator image (global scratch, arrival counter) and runs the epilogue. Sums are identical
    // whichever arrives last (a + b == b + a). The decision is taken on the device from the live row count.
    const int ntiles_all = (n + TM - 1) / TM;
    const bool split = a.partial != nullptr && kvol >= 8 && ntiles_all < a.split_below;
    // partner workgroups are the second half of the grid: consecutive block ids go round-robin over the 8 XCDs, so pairing
    // (2t, 2t+1) would park every unsplit launch on four of them
    const int grid_tiles = a.partial != nullptr ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int tile = (int)blockIdx.x % grid_tiles;
    const int half = (int)blockIdx.x / grid_tiles;
    if (!split && half) return;
    const int row0 = tile * TM;
    if (row0 >= n) return;
    const long long tstart = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    const auto rsrc_f = __builtin_amdgcn_make_buffer_rsrc((void *)a.feats, 0, (int)a.feats_bytes, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.weight, 0, kvol * COUT * CIN * 2, 0x00020000);

    if (tid == 0) mask_s[0] = 0;
    {   // gather-table slice of this tile: all loads in flight at once (bounded buffer loads), then one pass of LDS stores
        const auto rsrc_n = __builtin_amdgcn_make_buffer_rsrc((void *)(a.nbr + (size_t)row0 * kvol), 0, (n - row0 < TM ? n - row0 : TM) * kvol * 4, 0x00020000);
        int v[(TM * 27 + 255) / 256];
#pragma unroll
        for (int i = 0; i < (TM * 27 + 255) / 256; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_n, (tid + 256 * i) * 4, 0, 0) ;
        const int lim = (n - row0 < TM ? n - row0 : TM) * kvol;
#pragma unroll
        for (int i = 0; i < (TM * 27 + 255) / 256; ++i) {
            const int t = tid + 256 * i;
            if (t < TM * kvol) nbr_s[t] = t < lim ? v[i] : -1;
        }
    }
    __syncthreads();
    if (tid < TM) {
        int bits = 0;
        for (int k = 0; k < kvol; ++k)
            if (nbr_s[tid * kvol + k] >= 0) bits |= 1 << k;
        if (bits) atomicOr(&mask_s[0], bits);
    }
    __syncthreads();
    int live = mask_s[0];
    if (split) {                                        // this workgroup's share: the lower / upper half of the live offsets
        const int want = (__popc(live) + 1) >> 1;
        int low = 0, cnt = 0;
        for (int k = 0; k < kvol && cnt < want; ++k)
            if ((live >> k) & 1) { low |= 1 << k; ++cnt; }
        live = half ? (live & ~low) : low;
    }

    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // unit = (kernel offset k, channel half h): A fragments of this wave's 2 row tiles, weight half-slice W_k[:, h*UC .. +UC).
    // Two register sets (A0 / A1) ping-pong between "feeding the MFMAs" and "being filled for the next unit"; the loop body
    // holds one unit of each parity, so there are no register copies and the LDS buffer index is a constant. Everything in
    // a unit is branch-free: past the last unit (k = kvol) every load offset is out of range, i.e. zeros and no traffic, so
    // an odd unit count is simply padded with one null unit.
    i32x4v A0[RT][2][KC], A1[RT][2][KC], wn[NWL];
    auto loadW = [&](int k, int h) {
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int t = tid + 256 * i;
            const int c8 = t % (UC / 8), o = t / (UC / 8);
            const unsigned off = (WCH % 256 == 0 || t < WCH) ? (unsigned)(((k * COUT + o) * CIN + h * UC + c8 * 8) * 2) : WD_OOB;
            wn[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, off, 0, 0);       // k = kvol is past the end of the tensor
        }
    };
    auto storeW = [&](int b) {
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int t = tid + 256 * i;
            if (WCH % 256 == 0 || t < WCH) {
                const int c8 = t % (UC / 8), o = t / (UC / 8);
                *reinterpret_cast<i32x4v *>(Wb + (b * COUT + o) * LDW + (SWZ ? (c8 ^ ((o >> 1) & 7)) : c8) * 8) = wn[i];
            }
        }
    };
    // gather-table entries of this lane's RT rows for offset k (-1 past the last unit): read early, used by loadA later
    auto readSrc = [&](int (&src)[RT], int k) {
        const int kk = k < kvol ? k : 0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int v = nbr_s[(wave * (RT * 16) + rt * 16 + l15) * kvol + kk];
            src[rt] = k < kvol ? v : -1;
        }
    };
    auto loadA1 = [&](i32x4v (&A)[RT][2][KC], const int (&src)[RT], int h, int rt) {      // the 2*KC fragments of row tile rt
        const unsigned base = (unsigned)src[rt] * (unsigned)(2 * CIN * 2) + (unsigned)((h * UC + 8 * g) * 2);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const unsigned off = src[rt] >= 0 ? base + (unsigned)((half * CIN + kc * 32) * 2) : WD_OOB;
                A[rt][half][kc] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_f, off, 0, 0);
            }
    };
    auto loadA = [&](i32x4v (&A)[RT][2][KC], int k, int h) {
        int src[RT];
        readSrc(src, k);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) loadA1(A, src, h, rt);
    };
    auto next_unit = [&](int &k, int &h) {            // advance (k, h) to the next unit with work; k = kvol at the end
        if (k >= kvol) return;
        if (h + 1 < NU) { ++h; return; }
        h = 0;
        const int rest = live & ~((2 << k) - 1);
        k = rest ? (__ffs(rest) - 1) : kvol;
    };
    const f16x8 lo_scale = {(_Float16)(1.0f / 2048.f), (_Float16)(1.0f / 2048.f), (_Float16)(1.0f / 2048.f), (_Float16)(1.0f / 2048.f),
                            (_Float16)(1.0f / 2048.f), (_Float16)(1.0f / 2048.f), (_Float16)(1.0f / 2048.f), (_Float16)(1.0f / 2048.f)};
    long long st0 = 0, st1 = 0, st2 = 0, stn = 0, stf = 0;
    // One unit: MFMAs on Acur / LDS buffer b, with the next unit's traffic issued between the MFMA groups.
    auto unit = [&](i32x4v (&Acur)[RT][2][KC], i32x4v (&Anext)[RT][2][KC], int b, int &k, int &h) {
        const long long t0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        next_unit(k, h);                                 // (k, h) now names the NEXT unit
        loadW(k, h);                                     // oldest in the vm queue: its wait does not cover the A loads below
        f16x8 F[RT][2][KC];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                F[rt][0][kc] = __builtin_bit_cast(f16x8, Acur[rt][0][kc]);
                F[rt][1][kc] = __builtin_bit_cast(f16x8, Acur[rt][1][kc]) * lo_scale;   // lo/2048: exact power-of-two scale (subnormals kept)
            }
        // Weight fragments stream LDS -> registers three deep (an LDS read takes ~130-200 cycles, 4 MFMAs only 64), each one
        // feeds 4 MFMAs (2 row tiles x hi / lo). The scheduling fences keep this order; the next unit's gathers are issued
        // a quarter of the way in so that they land under the remaining MFMAs, the barrier and the head of the next unit.
        const long long tf = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        if constexpr (STAMP) stf += tf - t0;
        constexpr int NF = KC * NT;                      // fragments per unit, j = kc * NT + nt
        constexpr int FD = 4;                            // fragment reads in flight ahead of the MFMAs that use them
        const int c0 = SWZ ? (g ^ ((l15 >> 1) & 7)) : g;  // this lane's chunk of K chunk 0; K chunk 1 is c0 ^ 4 (swizzled) / c0 + 4
        const _Float16 *bb = Wb + (b * COUT + l15) * LDW;
        auto frag = [&](int j) {
            const int c = SWZ ? (c0 ^ ((j / NT) * 4)) : (c0 + (j / NT) * 4);
            return *reinterpret_cast<const f16x8 *>(bb + (j % NT) * 16 * LDW + c * 8);
        };
        f16x8 fb[FD + 1];
#pragma unroll
        for (int j = 0; j < FD && j < NF; ++j) fb[j] = frag(j);
        int src[RT];
        readSrc(src, k);                                 // the next unit's gather rows: their LDS latency hides under group 0
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (j + FD < NF) fb[(j + FD) % (FD + 1)] = frag(j + FD);
            if (j >= 1 && j <= RT) loadA1(Anext, src, h, j - 1);     // one row tile's gathers per MFMA group, not one long bubble
            __builtin_amdgcn_sched_barrier(0);
            const int nt = j % NT, kc = j / NT;
            const f16x8 f = fb[j % (FD + 1)];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F[rt][0][kc], f, acc[rt][nt], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F[rt][1][kc], f, acc[rt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const long long t1 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        storeW(b ^ 1);
        const long long t2 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        __syncthreads();                                 // W[b^1] visible; everyone is done reading W[b]
        if constexpr (STAMP) {
            const long long t3 = __builtin_amdgcn_s_memtime();
            st0 += t1 - t0; st1 += t2 - t1; st2 += t3 - t2; ++stn;
        }
    };

    int k = live ? (__ffs(live) - 1) : kvol, h = 0;
    const int nunits = __popc(live) * NU;
    loadW(k, h);
    loadA(A0, k, h);
    storeW(0);
    __syncthreads();
    const long long tloop = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    for (int u = 0; u < nunits; u += 2) {
        unit(A0, A1, 0, k, h);
        unit(A1, A0, 1, k, h);
    }
    if (STAMP && tid == 0) {
        atomicAdd((unsigned long long *)&a.stamps[0], (unsigned long long)st0);
        atomicAdd((unsigned long long *)&a.stamps[1], (unsigned long long)st1);
        atomicAdd((unsigned long long *)&a.stamps[2], (unsigned long long)st2);
        atomicAdd((unsigned long long *)&a.stamps[3], (unsigned long long)stn);
        atomicAdd((unsigned long long *)&a.stamps[4], (unsigned long long)(tloop - tstart));
        atomicAdd((unsigned long long *)&a.stamps[5], (unsigned long long)(__builtin_amdgcn_s_memtime() - tloop));
        atomicAdd((unsigned long long *)&a.stamps[6], 1ull);
        atomicAdd((unsigned long long *)&a.stamps[7], (unsigned long long)stf);
    }

    if (split) {
        // accumulator image: thread-major, 16 bytes per (thread, rt, nt) -> coalesced 16-byte accesses
        float *mine = a.partial + ((size_t)tile * 2 + half) * (TM * COUT);
        const float *other = a.partial + ((size_t)tile * 2 + (half ^ 1)) * (TM * COUT);
        __shared__ int s_arrival;
        if (tid == 0) s_arrival = __hip_atomic_load(&a.flags[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        int arrival = s_arrival;                        // 1: the partner has already published its image -> no need to write ours
        if (arrival == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    *reinterpret_cast<f32x4 *>(mine + ((size_t)(rt * NT + nt) * 256 + tid) * 4) = acc[rt][nt];
            __threadfence();
            __syncthreads();
            if (tid == 0) s_arrival = atomicAdd(&a.flags[tile], 1);
            __syncthreads();
            arrival = s_arrival;
            if (arrival == 0) return;                   // first to arrive: the partner finishes the tile
        }
        __threadfence();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 o = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(other + ((size_t)(rt * NT + nt) * 256 + tid) * 4));
                acc[rt][nt] = half ? (f32x4){__fadd_rn(o.x, acc[rt][nt].x), __fadd_rn(o.y, acc[rt][nt].y), __fadd_rn(o.z, acc[rt][nt].z), __fadd_rn(o.w, acc[rt][nt].w)}
                                   : (f32x4){__fadd_rn(acc[rt][nt].x, o.x), __fadd_rn(acc[rt][nt].y, o.y), __fadd_rn(acc[rt][nt].z, o.z), __fadd_rn(acc[rt][nt].w, o.w)};
            }
        if (tid == 0) a.flags[tile] = 0;                // ready for the next launch (stream-ordered)
    }

    // ---- epilogue (D layout: col = lane&15 -> out channel, row = 4g + reg -> tile row)
    float cmax[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int o = nt * 16 + l15;
        const float wdq = a.w_dequant[o];
        const float bs = a.bias ? a.bias[o] : 0.f;
        const float ps = a.post_scale ? a.post_scale[o] : 1.f;
        const float pt = a.post_shift ? a.post_shift[o] : 0.f;
        float mx = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = row0 + wave * (RT * 16) + rt * 16 + 4 * g + r;
                if (m < n) {
                    float v = __fmul_rn(acc[rt][nt][r], wdq);
                    if (a.bias) v = __fadd_rn(v, bs);
                    if (a.post_scale) v = __fadd_rn(__fmul_rn(v, ps), pt);
                    if (a.residual) v = __fadd_rn(v, a.residual[(size_t)m * COUT + o]);
                    if (a.relu) v = fmaxf(v, 0.f);
                    a.feats_out[(size_t)m * COUT + o] = v;
                    mx = fmaxf(mx, fabsf(v));
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        cmax[nt] = mx;
    }
    if (a.absmax_out) {
        if (g == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) red[wave * COUT + nt * 16 + l15] = cmax[nt];
        }
        __syncthreads();
        for (int o = tid; o < COUT; o += 256) {
            const float mx = fmaxf(fmaxf(red[o], red[COUT + o]), fmaxf(red[2 * COUT + o], red[3 * COUT + o]));
            if (mx > a.absmax_out[o]) aq3d_atomic_max_pos(&a.absmax_out[o], mx);
        }
    }
}

template <int CIN, int COUT>
static int launch_wide(const SpconvWideArgs &a_in, hipStream_t stream, bool stamp, void *ws, size_t ws_bytes) {
    SpconvWideArgs a = a_in;
    const int cap_tiles = aq3d_div_up(a.cap_out, 128);
    const size_t need = (((size_t)cap_tiles * 4 + 255) & ~(size_t)255) + (size_t)cap_tiles * 2 * 128 * COUT * 4;
    static const int split_below = getenv("AQ3D_WIDE_SPLIT_BELOW") ? atoi(getenv("AQ3D_WIDE_SPLIT_BELOW")) : 448;
    a.partial = nullptr; a.flags = nullptr; a.split_below = split_below;
    if (ws && ws_bytes >= need && split_below > 0 && a.kvol >= 8 && !stamp) {
        a.flags = (int *)ws;                                                   // flags first: zeroed once by the owner
        a.partial = (float *)((char *)ws + (((size_t)cap_tiles * 4 + 255) & ~(size_t)255));
    }
    constexpr int UC = CIN < 64 ? CIN : 64;
    constexpr size_t lds = (size_t)2 * COUT * (UC == 64 ? UC : UC + 8) * 2 + (size_t)128 * 27 * 4 + (size_t)4 * COUT * 4 + 64;
    const int grid = aq3d_div_up(a.cap_out, 128) * (a.partial ? 2 : 1);
    auto kern = spconv_wide_kernel<CIN, COUT, false>;
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (stamp) {          // diagnostic path only (instrumented instantiation): blocking, prints the stamp averages of this launch
        auto skern = spconv_wide_kernel<CIN, COUT, true>;
        if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(skern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        static long long *dev = nullptr;
        if (!dev) (void)hipMalloc((void **)&dev, 64);
        (void)hipMemsetAsync(dev, 0, 64, stream);
        SpconvWideArgs b = a;
        b.stamps = dev;
        hipLaunchKernelGGL(skern, dim3(grid < 1 ? 1 : grid), dim3(256), lds, stream, b);
        long long h[8];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, dev, 64, hipMemcpyDeviceToHost);
        const double nu = (double)(h[3] ? h[3] : 1), nw = (double)(h[6] ? h[6] : 1);
        fprintf(stderr, "[wide %dx%d] units/wg %.1f  cycles/unit: A-wait+scale %.0f  mfma-loop %.0f  storeW %.0f  barrier %.0f | per wg: prologue %.0f  loop %.0f\n",
                CIN, COUT, nu / nw, h[7] / nu, (h[0] - h[7]) / nu, h[1] / nu, h[2] / nu, h[4] / nw, h[5] / nw);
        return AQ3D_OK;
    }
    hipLaunchKernelGGL(kern, dim3(grid < 1 ? 1 : grid), dim3(256), lds, stream, a);
    AQ3D_CHECK_LAUNCH("aq3d_spconv(wide tile)");
    return AQ3D_OK;
}

// AQ3D_ERR_UNSUPPORTED when the shape has no wide-tile variant (caller falls back to spconv_split.hip)
int aq3d_spconv_wide_dispatch(const void *feats_split, unsigned feats_bytes, const void *weight_f16, const float *w_dequant,
                              const float *bias, const float *post_scale, const float *post_shift, const float *residual,
                              const int *nbr, const int *count_out, int cap_out, int cin, int cout, int kvol, int relu,
                              float *feats_out, float *absmax_out, void *ws, size_t ws_bytes, hipStream_t stream) {
    static const bool off = getenv("AQ3D_SPCONV_WIDE") != nullptr && atoi(getenv("AQ3D_SPCONV_WIDE")) == 0;
    if (off || kvol > 27) return AQ3D_ERR_UNSUPPORTED;
    SpconvWideArgs a;
    a.feats = (const _Float16 *)feats_split; a.weight = (const _Float16 *)weight_f16; a.w_dequant = w_dequant; a.bias = bias;
    a.post_scale = post_scale; a.post_shift = post_shift; a.residual = residual; a.nbr = nbr; a.count_out = count_out;
    a.feats_out = feats_out; a.absmax_out = absmax_out; a.cap_out = cap_out; a.kvol = kvol; a.relu = relu;
    a.feats_bytes = feats_bytes;
    static const bool stamp = getenv("AQ3D_WIDE_DEBUG") != nullptr && atoi(getenv("AQ3D_WIDE_DEBUG")) != 0;
    a.stamps = nullptr;
    if (cin == 32 && cout == 64) return launch_wide<32, 64>(a, stream, stamp, ws, ws_bytes);
    if (cin == 64 && cout == 64) return launch_wide<64, 64>(a, stream, stamp, ws, ws_bytes);
    if (cin == 64 && cout == 128) return launch_wide<64, 128>(a, stream, stamp, ws, ws_bytes);
    if (cin == 128 && cout == 128) return launch_wide<128, 128>(a, stream, stamp, ws, ws_bytes);
    return AQ3D_ERR_UNSUPPORTED;
}
