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
};

template <int CIN, int COUT, int WR, int WC, int G>
__global__ __launch_bounds__(256) void spconv_split_kernel(SpconvSplitArgs a) {
    constexpr int TM = 16 * WR;
    constexpr int NT = COUT / (16 * WC);
    constexpr int KP = ((G * CIN + 31) / 32) * 32;
    constexpr int LD = KP + 8;                         // halves per LDS row
    constexpr int C8 = CIN / 8;                        // 16-byte chunks per (row, offset, half)
    constexpr int AITEMS = TM * G * C8 * 2;            // hi and lo chunks
    constexpr int WITEMS = COUT * G * C8;
    constexpr int NA = (AITEMS + 255) / 256;
    constexpr int NW = (WITEMS + 255) / 256;
    static_assert(WR * WC == 4, "4 waves");
    static_assert(CIN % 16 == 0 && COUT % (16 * WC) == 0, "channel granularity");
    static_assert(NA <= 16 && NW <= 16, "okbits layout");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16 *Ahi = reinterpret_cast<_Float16 *>(smem_raw);   // [TM][LD]
    _Float16 *Alo = Ahi + TM * LD;                              // [TM][LD]
    _Float16 *Bs = Alo + TM * LD;                               // [COUT][LD]
    int *nbr_s = reinterpret_cast<int *>(Bs + COUT * LD);       // [TM][27]
    float *red = reinterpret_cast<float *>(nbr_s + TM * 27);    // [WR][COUT]
    int *mask_s = reinterpret_cast<int *>(red + WR * COUT);     // [0]: phases with work; [1+wr]: per 16-row block

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WR, wc = wave / WR;
    const int g = lane >> 4, l15 = lane & 15;
    const int kvol = a.kvol;
    const int nphase = (kvol + G - 1) / G;
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const int ntiles = (n + TM - 1) / TM;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * TM;
        __syncthreads();
        for (int t = tid; t < TM * kvol; t += 256) {
            int r = row0 + t / kvol;
            nbr_s[t] = (r < n) ? a.nbr[(size_t)row0 * kvol + t] : -1;
        }
        if (tid < 1 + WR) mask_s[tid] = 0;
        f32x4 acc_hi[NT], acc_lo[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            acc_hi[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc_lo[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        if (tid < TM) {
            int bits = 0;
            for (int k = 0; k < kvol; ++k)
                if (nbr_s[tid * kvol + k] >= 0) bits |= 1 << (k / G);
            if (bits) {
                atomicOr(&mask_s[0], bits);
                atomicOr(&mask_s[1 + (tid >> 4)], bits);
            }
        }
        __syncthreads();
        const int live = mask_s[0];
        const int mine = mask_s[1 + wr];

        int4 aq[NA], wq[NW], aq2[NA], wq2[NW];   // two phases of global loads in flight (the loop is L2-latency bound)
        // item t of the A stage: chunk c8 of half h (0 = hi, 1 = lo) of offset j of tile row r
        // Branch-free on purpose: every load is issued unconditionally (missing neighbours read row 0 and are zeroed by a
        // select) so that the instruction stream is straight-line and the compiler can use counted s_waitcnt vmcnt(N)
        // instead of draining the queue; that is what keeps two phases of loads in flight.
        auto prefetch = [&](int p, int4 (&aq)[NA], int4 (&wq)[NW], unsigned &okbits) {
            const int k0 = p * G, gn = min(G, kvol - k0);
            unsigned ob = 0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int t = (AITEMS % 256 == 0) ? tid + 256 * i : min(tid + 256 * i, AITEMS - 1);
                const int c8 = t % C8, h = (t / C8) & 1, j = (t / (2 * C8)) % G, r = t / (2 * C8 * G);
                const int jj = min(j, gn - 1);
                const int src = nbr_s[r * kvol + k0 + jj];
                ob |= ((src >= 0) & (j < gn)) ? (1u << i) : 0u;
                aq[i] = *reinterpret_cast<const int4 *>(a.feats + (size_t)max(src, 0) * (2 * CIN) + h * CIN + c8 * 8);
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int t = (WITEMS % 256 == 0) ? tid + 256 * i : min(tid + 256 * i, WITEMS - 1);
                const int c8 = t % C8, j = (t / C8) % G, o = t / (C8 * G);
                const int jj = min(j, gn - 1);
                if (G > 1) ob |= (j < gn) ? (1u << (16 + i)) : 0u;
                wq[i] = *reinterpret_cast<const int4 *>(a.weight + ((size_t)(k0 + jj) * COUT + o) * CIN + c8 * 8);
            }
            okbits = ob;
        };
        // the zero-select of missing neighbours happens here, when the data has arrived anyway
        auto store = [&](int4 (&aq)[NA], int4 (&wq)[NW], unsigned okbits) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int t = tid + 256 * i;
                if (AITEMS % 256 == 0 || t < AITEMS) {
                    const int c8 = t % C8, h = (t / C8) & 1, j = (t / (2 * C8)) % G, r = t / (2 * C8 * G);
                    const bool ok = (okbits >> i) & 1u;
                    int4 v = aq[i];
                    v.x = ok ? v.x : 0; v.y = ok ? v.y : 0; v.z = ok ? v.z : 0; v.w = ok ? v.w : 0;
                    *reinterpret_cast<int4 *>(Ahi + h * (TM * LD) + r * LD + j * CIN + c8 * 8) = v;
                }
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int t = tid + 256 * i;
                if (WITEMS % 256 == 0 || t < WITEMS) {
                    const int c8 = t % C8, j = (t / C8) % G, o = t / (C8 * G);
                    int4 v = wq[i];
                    if (G > 1) { const bool ok = (okbits >> (16 + i)) & 1u; v.x = ok ? v.x : 0; v.y = ok ? v.y : 0; v.z = ok ? v.z : 0; v.w = ok ? v.w : 0; }
                    *reinterpret_cast<int4 *>(Bs + o * LD + j * CIN + c8 * 8) = v;
                }
            }
        };
        if constexpr (KP > G * CIN) {
            constexpr int PADC = KP - G * CIN;
            for (int t = tid; t < (2 * TM + COUT) * PADC; t += 256) {
                const int r = t / PADC, c = G * CIN + t % PADC;
                Ahi[r * LD + c] = (_Float16)0.f;     // Ahi, Alo and Bs are contiguous with the same row pitch
            }
        }

        auto next_live = [&](int p) -> int {
            if (p >= nphase) return nphase;
            const int rest = live & ~((2 << p) - 1);
            return rest ? (__ffs(rest) - 1) : nphase;
        };
        int p = live ? (__ffs(live) - 1) : nphase;
        int pn = next_live(p);
        unsigned ok1 = 0, ok2 = 0;
        if (p < nphase) prefetch(p, aq, wq, ok1);
        if (pn < nphase) prefetch(pn, aq2, wq2, ok2);
        // one pipeline step: the staged registers of phase p go to LDS, the freed registers immediately take the loads
        // of the phase after next, then the MFMAs of phase p run under two phases of loads in flight.
        auto step = [&](int4 (&sa)[NA], int4 (&sw)[NW], unsigned &sok) {
            __syncthreads();
            store(sa, sw, sok);
            __syncthreads();
            const int pnn = next_live(pn);
            if (pnn < nphase) prefetch(pnn, sa, sw, sok);
            if ((mine >> p) & 1) {
                const _Float16 *ah = Ahi + (wr * 16 + l15) * LD + 8 * g;
                const _Float16 *al = Alo + (wr * 16 + l15) * LD + 8 * g;
                const _Float16 *bb = Bs + ((wc * NT) * 16 + l15) * LD + 8 * g;
                // full phase K every time (offsets beyond kvol were stored as zeros): a compile-time trip count lets the
                // compiler hoist the LDS fragment reads of later chunks above the MFMAs of earlier ones
#pragma unroll
                for (int kc = 0; kc < KP; kc += 32) {
                    const f16x8 fh = *reinterpret_cast<const f16x8 *>(ah + kc);
                    const f16x8 fl = *reinterpret_cast<const f16x8 *>(al + kc);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const f16x8 fb = *reinterpret_cast<const f16x8 *>(bb + nt * 16 * LD + kc);
                        acc_hi[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, fb, acc_hi[nt], 0, 0, 0);
                        acc_lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, fb, acc_lo[nt], 0, 0, 0);
                    }
                }
            }
            p = pn;
            pn = pnn;
        };
        while (p < nphase) {
            step(aq, wq, ok1);
            if (p >= nphase) break;
            step(aq2, wq2, ok2);
        }

        // ---- epilogue (D layout: col = lane&15 → out channel, row = 4g + reg → tile row)
        float cmax[NT];
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
                const int m = row0 + wr * 16 + 4 * g + r;
                if (m < n) {
                    float v = __fadd_rn(acc_hi[nt][r], __fmul_rn(acc_lo[nt][r], 1.0f / 2048.f));
                    v = __fmul_rn(v, wdq);
                    if (a.bias) v = __fadd_rn(v, b);
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
                for (int nt = 0; nt < NT; ++nt) red[wr * COUT + (wc * NT + nt) * 16 + l15] = cmax[nt];
            }
            __syncthreads();
            for (int o = tid; o < COUT; o += 256) {
                float mx = red[o];
#pragma unroll
                for (int w = 1; w < WR; ++w) mx = fmaxf(mx, red[w * COUT + o]);
                if (mx > a.absmax_out[o]) aq3d_atomic_max_pos(&a.absmax_out[o], mx);
            }
        }
    }
}

template <int CIN, int COUT, int WR, int WC, int G>
static int launch_split(const SpconvSplitArgs &a, hipStream_t stream) {
    constexpr int TM = 16 * WR;
    constexpr int KP = ((G * CIN + 31) / 32) * 32;
    constexpr size_t lds = (size_t)(2 * TM + COUT) * (KP + 8) * 2 + (size_t)TM * 27 * 4 + (size_t)WR * COUT * 4 + 32;
    const int tiles = aq3d_div_up(a.cap_out, TM);
    const int grid = tiles < 4096 ? (tiles < 1 ? 1 : tiles) : 4096;
    auto kern = spconv_split_kernel<CIN, COUT, WR, WC, G>;
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    AQ3D_CHECK_LAUNCH("aq3d_spconv(f16 split)");
    return AQ3D_OK;
}

int aq3d_spconv_split_dispatch(const void *feats_split, const void *weight_f16, const float *w_dequant, const float *bias,
                               const float *post_scale, const float *post_shift, const float *residual, const int *nbr,
                               const int *count_out, int cap_out, int cin, int cout, int kvol, int relu, float *feats_out,
                               float *absmax_out, hipStream_t stream) {
    SpconvSplitArgs a;
    a.feats = (const _Float16 *)feats_split; a.weight = (const _Float16 *)weight_f16; a.w_dequant = w_dequant; a.bias = bias;
    a.post_scale = post_scale; a.post_shift = post_shift; a.residual = residual; a.nbr = nbr; a.count_out = count_out;
    a.feats_out = feats_out; a.absmax_out = absmax_out; a.cap_out = cap_out; a.kvol = kvol; a.relu = relu;
#define AQ3D_SPLIT_CASE(CI, CO, WR_, WC_, G_) \
    if (cin == CI && cout == CO) return launch_split<CI, CO, WR_, WC_, G_>(a, stream);
    AQ3D_SPLIT_CASE(16, 16, 4, 1, 9)
    AQ3D_SPLIT_CASE(16, 32, 2, 2, 9)
    AQ3D_SPLIT_CASE(32, 32, 2, 2, 4)
    AQ3D_SPLIT_CASE(32, 64, 2, 2, 4)
    AQ3D_SPLIT_CASE(64, 64, 2, 2, 2)
    AQ3D_SPLIT_CASE(64, 128, 2, 2, 2)
    AQ3D_SPLIT_CASE(128, 128, 2, 2, 1)
#undef AQ3D_SPLIT_CASE
    aq3d_set_error("aq3d_spconv: no f16-split variant for cin=%d cout=%d", cin, cout);
    return AQ3D_ERR_UNSUPPORTED;
}
