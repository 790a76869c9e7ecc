// Sparse conv on pre-split fp16 activations x fp16-held integer weight codes: the production path of the QConvNd layers.
//
// aq3d_quantize_rows already touched every activation once and wrote its fake-quantised value v = q * d_c as an fp16
// pair (hi, lo) with v = hi + lo/2048 (22 bits). A feature row is gathered by ~8-14 outputs, so doing the scale + split
// there (once per element) instead of in the gather (once per (output, tap)) removes ~90 % of the VALU work the
// int8-coded variant (spconv_i8.hip) spent converting, and turns the gather into pure 16-byte copies of 512-byte rows.
// Weights are the per-out-channel int8 codes held as fp16 (exact), so their staging is a copy too. Every
// v_mfma_f32_16x16x32_f16 product is exact in fp32; acc = sum(hi*w) + sum(lo*w)/2048, weight scale in the epilogue.
//
// Tiling as in spconv_i8.hip: workgroup = TM rows x all COUT columns, kernel offsets in phases of G (K per phase = G*CIN),
// register prefetch of the next live phase under the MFMAs, per-tile phase masks, 16-row blocks skip empty phases.
#include "common.h"
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SpconvSplitArgs {
    const _Float16 *feats;    // [M_in, 2*CIN]  row = hi(CIN) | lo(CIN)
    const _Float16 *weight;   // [kvol][COUT][CIN] integer codes as fp16
    const float *w_dequant;   // [COUT]
    const float *bias, *post_scale, *post_shift, *residual;
    const int *nbr;           // [cap_out][kvol]
    const int *count_out;
    float *feats_out, *absmax_out;
    int cap_out, kvol, relu;
    unsigned feats_bytes;     // capacity of the feature buffer (buffer-load bounds: out-of-range reads return 0)
};

typedef int i32x4v __attribute__((ext_vector_type(4)));
#define SPLIT_OOB 0xfffffff0u

// Workgroup = NSUB sub-tiles of TM = 16*WR rows, all COUT columns. Per phase (G kernel offsets) the weight slice is
// staged ONCE and reused by the NSUB sub-tiles (their accumulators stay in registers), which divides the dominant L2
// traffic — re-streaming the layer's weights per row tile — by NSUB. All global reads are buffer loads: a missing
// neighbour or an offset beyond kvol is an out-of-range offset, which the hardware answers with zeros without touching
// memory — no branches, no dummy traffic, straight-line code with counted vmcnt waits.
template <int CIN, int COUT, int WR, int WC, int G, int NSUB>
__global__ __launch_bounds__(64 * WR * WC) void spconv_split_kernel(SpconvSplitArgs a) {
    constexpr int NTH = 64 * WR * WC;                   // threads per workgroup
    constexpr int TM = 16 * WR;
    constexpr int TMS = TM * NSUB;
    constexpr int NT = COUT / (16 * WC);
    constexpr int KP = ((G * CIN + 31) / 32) * 32;
    constexpr int LD = KP + 8;                         // halves per LDS row
    constexpr int C8 = CIN / 8;                        // 16-byte chunks per (row, offset, half)
    constexpr int AITEMS = TM * G * C8 * 2;            // hi and lo chunks of one sub-tile
    constexpr int WITEMS = COUT * G * C8;
    constexpr int NA = (AITEMS + NTH - 1) / NTH;
    constexpr int NW = (WITEMS + NTH - 1) / NTH;
    static_assert(WR * WC == 4 || WR * WC == 8, "4 or 8 waves");
    static_assert(CIN % 16 == 0 && COUT % (16 * WC) == 0, "channel granularity");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16 *Ahi = reinterpret_cast<_Float16 *>(smem_raw);   // [TM][LD]
    _Float16 *Alo = Ahi + TM * LD;                              // [TM][LD]
    _Float16 *Bs = Alo + TM * LD;                               // [COUT][LD]
    int *nbr_s = reinterpret_cast<int *>(Bs + COUT * LD);       // [TMS][27]
    float *red = reinterpret_cast<float *>(nbr_s + TMS * 27);   // [WR][COUT]
    int *mask_s = reinterpret_cast<int *>(red + WR * COUT);     // [u]: phases with work in sub-tile u; [NSUB + u*WR + wr]: per 16-row block

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WR, wc = wave / WR;
    const int g = lane >> 4, l15 = lane & 15;
    const int kvol = a.kvol;
    const int nphase = (kvol + G - 1) / G;
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const int ntiles = (n + TMS - 1) / TMS;
    const auto rsrc_f = __builtin_amdgcn_make_buffer_rsrc((void *)a.feats, 0, (int)a.feats_bytes, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.weight, 0, kvol * COUT * CIN * 2, 0x00020000);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * TMS;
        __syncthreads();
        for (int t = tid; t < TMS * kvol; t += NTH) {
            int r = row0 + t / kvol;
            nbr_s[t] = (r < n) ? a.nbr[(size_t)row0 * kvol + t] : -1;
        }
        if (tid < NSUB + NSUB * WR) mask_s[tid] = 0;
        f32x4 acc_hi[NSUB][NT], acc_lo[NSUB][NT];
#pragma unroll
        for (int u = 0; u < NSUB; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                acc_hi[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc_lo[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        __syncthreads();
        for (int r = tid; r < TMS; r += NTH) {
            int bits = 0;
            for (int k = 0; k < kvol; ++k)
                if (nbr_s[r * kvol + k] >= 0) bits |= 1 << (k / G);
            if (bits) {
                atomicOr(&mask_s[r / TM], bits);
                atomicOr(&mask_s[NSUB + (r / TM) * WR + ((r % TM) >> 4)], bits);
            }
        }
        __syncthreads();
        int submask[NSUB], mine[NSUB];
        int live = 0;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            submask[u] = mask_s[u];
            mine[u] = mask_s[NSUB + u * WR + wr];
            live |= submask[u];
        }

        i32x4v aq[NA], wq[NW];
        auto prefetchA = [&](int p, int u) {
            const int k0 = p * G, gn = min(G, kvol - k0);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int t = (AITEMS % NTH == 0) ? tid + NTH * i : min(tid + NTH * i, AITEMS - 1);
                const int c8 = t % C8, h = (t / C8) & 1, j = (t / (2 * C8)) % G, r = t / (2 * C8 * G);
                const int src = nbr_s[(u * TM + r) * kvol + k0 + min(j, gn - 1)];
                const unsigned off = (src >= 0 && j < gn) ? (unsigned)((src * 2 * CIN + h * CIN + c8 * 8) * 2) : SPLIT_OOB;
                aq[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_f, off, 0, 0);
            }
        };
        auto prefetchB = [&](int p) {
            const int k0 = p * G, gn = min(G, kvol - k0);
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int t = (WITEMS % NTH == 0) ? tid + NTH * i : min(tid + NTH * i, WITEMS - 1);
                const int c8 = t % C8, j = (t / C8) % G, o = t / (C8 * G);
                const unsigned off = (j < gn) ? (unsigned)((((k0 + j) * COUT + o) * CIN + c8 * 8) * 2) : SPLIT_OOB;
                wq[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, off, 0, 0);
            }
        };
        auto storeA = [&]() {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int t = tid + NTH * i;
                if (AITEMS % NTH == 0 || t < AITEMS) {
                    const int c8 = t % C8, h = (t / C8) & 1, j = (t / (2 * C8)) % G, r = t / (2 * C8 * G);
                    *reinterpret_cast<i32x4v *>(Ahi + h * (TM * LD) + r * LD + j * CIN + c8 * 8) = aq[i];
                }
            }
        };
        auto storeB = [&]() {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int t = tid + NTH * i;
                if (WITEMS % NTH == 0 || t < WITEMS) {
                    const int c8 = t % C8, j = (t / C8) % G, o = t / (C8 * G);
                    *reinterpret_cast<i32x4v *>(Bs + o * LD + j * CIN + c8 * 8) = wq[i];
                }
            }
        };
        if constexpr (KP > G * CIN) {
            constexpr int PADC = KP - G * CIN;
            for (int t = tid; t < (2 * TM + COUT) * PADC; t += NTH) {
                const int r = t / PADC, c = G * CIN + t % PADC;
                Ahi[r * LD + c] = (_Float16)0.f;     // Ahi, Alo and Bs are contiguous with the same row pitch
            }
        }
        auto next_live = [&](int p) -> int {
            if (p >= nphase) return nphase;
            const int rest = live & ~((2 << p) - 1);
            return rest ? (__ffs(rest) - 1) : nphase;
        };
        auto first_sub = [&](int p, int ufrom) -> int {
            int res = NSUB;
#pragma unroll
            for (int u = NSUB - 1; u >= 0; --u)
                if (u >= ufrom && ((submask[u] >> p) & 1)) res = u;
            return res;
        };

        int p = live ? (__ffs(live) - 1) : nphase;
        if (p < nphase) {
            prefetchB(p);
            prefetchA(p, first_sub(p, 0));
        }
        while (p < nphase) {
            bool newphase = true;
#pragma unroll
            for (int u = 0; u < NSUB; ++u) {
                if (!((submask[u] >> p) & 1)) continue;            // block-uniform
                __syncthreads();                                   // previous unit's MFMAs are done with the tiles
                if (newphase) { storeB(); newphase = false; }
                storeA();
                __syncthreads();
                int un = first_sub(p, u + 1), pn = p;
                if (un == NSUB) {
                    pn = next_live(p);
                    un = pn < nphase ? first_sub(pn, 0) : 0;
                }
                if (pn < nphase) {                                  // next unit's loads fly under this unit's MFMAs
                    prefetchA(pn, un);
                    if (pn != p) prefetchB(pn);
                }
                if ((mine[u] >> p) & 1) {
                    const _Float16 *ah = Ahi + (wr * 16 + l15) * LD + 8 * g;
                    const _Float16 *al = Alo + (wr * 16 + l15) * LD + 8 * g;
                    const _Float16 *bb = Bs + ((wc * NT) * 16 + l15) * LD + 8 * g;
#pragma unroll 2
                    for (int kc = 0; kc < KP; kc += 32) {
                        const f16x8 fh = *reinterpret_cast<const f16x8 *>(ah + kc);
                        const f16x8 fl = *reinterpret_cast<const f16x8 *>(al + kc);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const f16x8 fb = *reinterpret_cast<const f16x8 *>(bb + nt * 16 * LD + kc);
                            acc_hi[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, fb, acc_hi[u][nt], 0, 0, 0);
                            acc_lo[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, fb, acc_lo[u][nt], 0, 0, 0);
                        }
                    }
                }
            }
            p = next_live(p);
        }

        // ---- epilogue (D layout: col = lane&15 → out channel, row = 4g + reg → tile row)
        float cmax[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) cmax[nt] = 0.f;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int o = (wc * NT + nt) * 16 + l15;
                const float wdq = a.w_dequant[o];
                const float b = a.bias ? a.bias[o] : 0.f;
                const float ps = a.post_scale ? a.post_scale[o] : 1.f;
                const float pt = a.post_shift ? a.post_shift[o] : 0.f;
                float mx = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = row0 + u * TM + wr * 16 + 4 * g + r;
                    if (m < n) {
                        float v = __fadd_rn(acc_hi[u][nt][r], __fmul_rn(acc_lo[u][nt][r], 1.0f / 2048.f));
                        v = __fmul_rn(v, wdq);
                        if (a.bias) v = __fadd_rn(v, b);
                        if (a.post_scale) v = __fadd_rn(__fmul_rn(v, ps), pt);
                        if (a.residual) v = __fadd_rn(v, a.residual[(size_t)m * COUT + o]);
                        if (a.relu) v = fmaxf(v, 0.f);
                        a.feats_out[(size_t)m * COUT + o] = v;
                        mx = fmaxf(mx, fabsf(v));
                    }
                }
                cmax[nt] = fmaxf(cmax[nt], mx);
            }
        }
        if (a.absmax_out) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float mx = cmax[nt];
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (g == 0) red[wr * COUT + (wc * NT + nt) * 16 + l15] = mx;
            }
            __syncthreads();
            for (int o = tid; o < COUT; o += NTH) {
                float mx = red[o];
#pragma unroll
                for (int w = 1; w < WR; ++w) mx = fmaxf(mx, red[w * COUT + o]);
                if (mx > a.absmax_out[o]) aq3d_atomic_max_pos(&a.absmax_out[o], mx);
            }
        }
    }
}

template <int CIN, int COUT, int WR, int WC, int G, int NSUB>
static int launch_split(const SpconvSplitArgs &a, hipStream_t stream) {
    constexpr int TM = 16 * WR;
    constexpr int KP = ((G * CIN + 31) / 32) * 32;
    constexpr size_t lds = (size_t)(2 * TM + COUT) * (KP + 8) * 2 + (size_t)TM * NSUB * 27 * 4 + (size_t)WR * COUT * 4 +
                           (size_t)(NSUB + NSUB * WR) * 4 + 32;
    const int tiles = aq3d_div_up(a.cap_out, TM * NSUB);
    const int grid = tiles < 4096 ? (tiles < 1 ? 1 : tiles) : 4096;
    auto kern = spconv_split_kernel<CIN, COUT, WR, WC, G, NSUB>;
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WR * WC), lds, stream, a);
    AQ3D_CHECK_LAUNCH("aq3d_spconv(f16 split)");
    return AQ3D_OK;
}

int aq3d_spconv_split_dispatch(const void *feats_split, unsigned feats_bytes, const void *weight_f16, const float *w_dequant,
                               const float *bias, const float *post_scale, const float *post_shift, const float *residual,
                               const int *nbr, const int *count_out, int cap_out, int cin, int cout, int kvol, int relu,
                               float *feats_out, float *absmax_out, hipStream_t stream) {
    SpconvSplitArgs a;
    a.feats = (const _Float16 *)feats_split; a.weight = (const _Float16 *)weight_f16; a.w_dequant = w_dequant; a.bias = bias;
    a.post_scale = post_scale; a.post_shift = post_shift; a.residual = residual; a.nbr = nbr; a.count_out = count_out;
    a.feats_out = feats_out; a.absmax_out = absmax_out; a.cap_out = cap_out; a.kvol = kvol; a.relu = relu;
    a.feats_bytes = feats_bytes;
    static const int nsub_env = getenv("AQ3D_SPCONV_NSUB") ? atoi(getenv("AQ3D_SPCONV_NSUB")) : 0;
#define AQ3D_SPLIT_CASE(CI, CO, WR_, WC_, G_, NS_DEFAULT)                                              \
    if (cin == CI && cout == CO) {                                                                     \
        const int ns = nsub_env ? nsub_env : NS_DEFAULT;                                               \
        if (ns >= 4) return launch_split<CI, CO, WR_, WC_, G_, 4>(a, stream);                          \
        if (ns >= 2) return launch_split<CI, CO, WR_, WC_, G_, 2>(a, stream);                          \
        return launch_split<CI, CO, WR_, WC_, G_, 1>(a, stream);                                       \
    }
    // measured on MI355X (W150k frame): sharing the weight slice across sub-tiles (NSUB 2/4) costs more in registers /
    // occupancy than it saves in L2 traffic, so every shape runs NSUB = 1.
    AQ3D_SPLIT_CASE(16, 16, 4, 1, 9, 1)
    AQ3D_SPLIT_CASE(16, 32, 2, 2, 9, 1)
    AQ3D_SPLIT_CASE(32, 32, 2, 2, 4, 1)
    AQ3D_SPLIT_CASE(32, 64, 2, 2, 4, 1)
    AQ3D_SPLIT_CASE(64, 64, 2, 2, 2, 1)
    AQ3D_SPLIT_CASE(64, 128, 2, 2, 2, 1)
    AQ3D_SPLIT_CASE(128, 128, 2, 2, 1, 1)
#undef AQ3D_SPLIT_CASE
    aq3d_set_error("aq3d_spconv: no f16-split variant for cin=%d cout=%d", cin, cout);
    return AQ3D_ERR_UNSUPPORTED;
}
