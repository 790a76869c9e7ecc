// Fast path of the sparse conv for int8-coded activations x int8-coded weights (the QConvNd layers).
//
// Why not the int8 MFMA: QConvNd quantises activations PER INPUT CHANNEL (quant/quant.py:21-26, axis=1), so the
// activation scale d_c sits inside the K contraction: sum_c (q_a[c] d_c)(q_w[o,c] t_o). An int8 x int8 dot cannot carry
// it. Instead the scale is folded into the gather (a = q_a * d_c, exactly the reference's fake-quantised value), and a
// is split into two fp16 numbers, a = hi + lo/2048 (hi = fp16(a), lo = fp16((a - hi) * 2048)), which carries 22
// significant bits. Weights are small integers and exact in fp16, so every product of v_mfma_f32_16x16x32_f16 is exact in
// fp32; two MFMAs (hi, lo) per K-chunk give fp32-grade accuracy at 8x the f32-MFMA rate. The weight scale t_o is applied
// in the epilogue. No per-frame weight preparation, and the weights travel from L2 as 1 byte each.
//
// Tiling: a 256-thread workgroup owns TM = 16*WR output rows x all COUT columns. The 27 kernel offsets are processed in
// PHASES of G offsets (K per phase KP = G*CIN = 128..144) so that each barrier pair covers a long K: one phase gathers
// G neighbour rows per output row (int8 codes → scaled fp16 hi/lo tiles in LDS) and stages G weight slices
// (int8 → fp16), then every wave runs KP/32 K-chunks x 2 MFMAs per 16x16 block. 16-row blocks without any neighbour in
// the phase skip their MFMAs (wave-uniform), whole phases without neighbours are skipped by the workgroup.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SpconvI8Args {
    const int8_t *feats_in;   // [M_in, CIN] codes
    const float *act_dequant; // [CIN]
    const int8_t *weight;     // [kvol][COUT][CIN] codes
    const float *w_dequant;   // [COUT]
    const float *bias, *post_scale, *post_shift, *residual;
    const int *nbr;           // [cap_out][kvol]
    const int *count_out;
    float *feats_out, *absmax_out;
    int cap_out, kvol, relu;
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// 4 packed int8 → 4 fp16 (exact) in 5 VALU ops: bias to unsigned, splice each byte under the fp16 exponent of 1024
// (0x64xx = 1024 + xx), subtract 1152 = 1024 + 128 with a packed add.
__device__ __forceinline__ void cvt4_i8_f16(int w, f16x2 &lo, f16x2 &hi) {
    const unsigned t = (unsigned)w ^ 0x80808080u;
    const unsigned p0 = __builtin_amdgcn_perm(t, 0x64646464u, 0x00050004u);   // bytes: t.b0, 0x64, t.b1, 0x64
    const unsigned p1 = __builtin_amdgcn_perm(t, 0x64646464u, 0x00070006u);   // bytes: t.b2, 0x64, t.b3, 0x64
    const f16x2 bias = {(_Float16)1152.f, (_Float16)1152.f};
    lo = __builtin_bit_cast(f16x2, p0) - bias;
    hi = __builtin_bit_cast(f16x2, p1) - bias;
}

__device__ __forceinline__ f16x8 cvt8_i8_f16(int w0, int w1) {
    f16x2 a, b, c, d;
    cvt4_i8_f16(w0, a, b);
    cvt4_i8_f16(w1, c, d);
    f16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1]; r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1];
    return r;
}

template <int CIN, int COUT, int WR, int WC, int G>
__global__ __launch_bounds__(256) void spconv_i8_kernel(SpconvI8Args a) {
    constexpr int TM = 16 * WR;
    constexpr int NT = COUT / (16 * WC);
    constexpr int KP = ((G * CIN + 31) / 32) * 32;   // phase K, padded to the MFMA K
    constexpr int LD = KP + 8;                       // halves per LDS row (16-byte pad: conflict-free b128 reads)
    constexpr int C16 = CIN / 16;
    constexpr int NA = (TM * G * C16 + 255) / 256;   // gather items (16 codes) per thread per phase
    constexpr int NW = (COUT * G * C16 + 255) / 256; // weight items (16 codes) per thread per phase
    static_assert(WR * WC == 4, "4 waves");
    static_assert(CIN % 16 == 0 && COUT % (16 * WC) == 0, "channel granularity");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16 *Ahi = reinterpret_cast<_Float16 *>(smem_raw);   // [TM][LD]
    _Float16 *Alo = Ahi + TM * LD;                              // [TM][LD]
    _Float16 *Bs = Alo + TM * LD;                               // [COUT][LD]
    int *nbr_s = reinterpret_cast<int *>(Bs + COUT * LD);       // [TM][27]
    float *red = reinterpret_cast<float *>(nbr_s + TM * 27);    // [WR][COUT]
    float *dq_s = red + WR * COUT;                              // [CIN]
    int *mask_s = reinterpret_cast<int *>(dq_s + CIN);          // [0]: phases with work; [1+wr]: per 16-row block

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WR, wc = wave / WR;
    const int g = lane >> 4, l15 = lane & 15;
    const int kvol = a.kvol;
    const int nphase = (kvol + G - 1) / G;
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const int ntiles = (n + TM - 1) / TM;
    for (int c = tid; c < CIN; c += 256) dq_s[c] = a.act_dequant[c];

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
        // phase masks: bit p set iff some row (of the tile / of a 16-row block) has a neighbour among phase p's offsets
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

        int4 aq[NA], wq[NW];
        auto prefetch = [&](int p) {
            const int k0 = p * G, gn = min(G, kvol - k0);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int t = tid + 256 * i;
                int4 v = make_int4(0, 0, 0, 0);
                if (t < TM * G * C16) {
                    const int c16 = t % C16, j = (t / C16) % G, r = t / (C16 * G);
                    const int src = (j < gn) ? nbr_s[r * kvol + k0 + j] : -1;
                    if (src >= 0) v = *reinterpret_cast<const int4 *>(a.feats_in + (size_t)src * CIN + c16 * 16);
                }
                aq[i] = v;
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int t = tid + 256 * i;
                int4 v = make_int4(0, 0, 0, 0);
                if (t < COUT * G * C16) {
                    const int c16 = t % C16, j = (t / C16) % G, o = t / (C16 * G);
                    if (j < gn) v = *reinterpret_cast<const int4 *>(a.weight + ((size_t)(k0 + j) * COUT + o) * CIN + c16 * 16);
                }
                wq[i] = v;
            }
        };
        auto store = [&]() {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int t = tid + 256 * i;
                if (t < TM * G * C16) {
                    const int c16 = t % C16, j = (t / C16) % G, r = t / (C16 * G);
                    const int w[4] = {aq[i].x, aq[i].y, aq[i].z, aq[i].w};
                    f16x8 h0, h1, l0, l1;
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const float q = (float)(int8_t)((w[u >> 2] >> (8 * (u & 3))) & 0xff);
                        const float x = __fmul_rn(q, dq_s[c16 * 16 + u]);       // the reference's fake-quantised value
                        const _Float16 h = (_Float16)x;
                        const _Float16 l = (_Float16)__fmul_rn(__fsub_rn(x, (float)h), 2048.f);
                        if (u < 8) { h0[u] = h; l0[u] = l; } else { h1[u - 8] = h; l1[u - 8] = l; }
                    }
                    _Float16 *dh = Ahi + r * LD + j * CIN + c16 * 16;
                    _Float16 *dl = Alo + r * LD + j * CIN + c16 * 16;
                    *reinterpret_cast<f16x8 *>(dh) = h0;
                    *reinterpret_cast<f16x8 *>(dh + 8) = h1;
                    *reinterpret_cast<f16x8 *>(dl) = l0;
                    *reinterpret_cast<f16x8 *>(dl + 8) = l1;
                }
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int t = tid + 256 * i;
                if (t < COUT * G * C16) {
                    const int c16 = t % C16, j = (t / C16) % G, o = t / (C16 * G);
                    _Float16 *dst = Bs + o * LD + j * CIN + c16 * 16;
                    *reinterpret_cast<f16x8 *>(dst) = cvt8_i8_f16(wq[i].x, wq[i].y);
                    *reinterpret_cast<f16x8 *>(dst + 8) = cvt8_i8_f16(wq[i].z, wq[i].w);
                }
            }
        };
        // the K padding columns (G*CIN not a multiple of 32) are written once per tile: zeros
        if constexpr (KP > G * CIN) {
            constexpr int PADC = KP - G * CIN;
            for (int t = tid; t < (2 * TM + COUT) * PADC; t += 256) {
                const int r = t / PADC, c = G * CIN + t % PADC;
                Ahi[r * LD + c] = (_Float16)0.f;     // Ahi, Alo and Bs are contiguous with the same row pitch
            }
        }

        int p = live ? (__ffs(live) - 1) : nphase;
        if (p < nphase) prefetch(p);
        while (p < nphase) {
            __syncthreads();            // every wave is done reading the previous phase's tiles
            store();
            __syncthreads();
            const int rest = live & ~((2 << p) - 1);
            const int pn = rest ? (__ffs(rest) - 1) : nphase;
            if (pn < nphase) prefetch(pn);  // global loads fly under the MFMAs below
            if ((mine >> p) & 1) {
                const int gn = min(G, kvol - p * G);
                const _Float16 *ah = Ahi + (wr * 16 + l15) * LD + 8 * g;
                const _Float16 *al = Alo + (wr * 16 + l15) * LD + 8 * g;
                const _Float16 *bb = Bs + ((wc * NT) * 16 + l15) * LD + 8 * g;
                const int kend = ((gn * CIN + 31) / 32) * 32;
#pragma unroll 2
                for (int kc = 0; kc < kend; kc += 32) {
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
static int launch_i8(const SpconvI8Args &a, hipStream_t stream) {
    constexpr int TM = 16 * WR;
    constexpr int KP = ((G * CIN + 31) / 32) * 32;
    constexpr size_t lds = (size_t)(2 * TM + COUT) * (KP + 8) * 2 + (size_t)TM * 27 * 4 + (size_t)WR * COUT * 4 + CIN * 4 + 32;
    const int tiles = aq3d_div_up(a.cap_out, TM);
    const int grid = tiles < 4096 ? (tiles < 1 ? 1 : tiles) : 4096;
    auto kern = spconv_i8_kernel<CIN, COUT, WR, WC, G>;
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    AQ3D_CHECK_LAUNCH("aq3d_spconv(i8)");
    return AQ3D_OK;
}

// returns AQ3D_ERR_UNSUPPORTED when there is no fast variant for (cin, cout)
int aq3d_spconv_i8_dispatch(const int8_t *feats_in, const float *act_dequant, const int8_t *weight, const float *w_dequant,
                            const float *bias, const float *post_scale, const float *post_shift, const float *residual,
                            const int *nbr, const int *count_out, int cap_out, int cin, int cout, int kvol, int relu,
                            float *feats_out, float *absmax_out, hipStream_t stream) {
    SpconvI8Args a;
    a.feats_in = feats_in; a.act_dequant = act_dequant; a.weight = weight; a.w_dequant = w_dequant; a.bias = bias;
    a.post_scale = post_scale; a.post_shift = post_shift; a.residual = residual; a.nbr = nbr; a.count_out = count_out;
    a.feats_out = feats_out; a.absmax_out = absmax_out; a.cap_out = cap_out; a.kvol = kvol; a.relu = relu;
#define AQ3D_I8_CASE(CI, CO, WR_, WC_, G_) \
    if (cin == CI && cout == CO) return launch_i8<CI, CO, WR_, WC_, G_>(a, stream);
    AQ3D_I8_CASE(16, 16, 4, 1, 9)
    AQ3D_I8_CASE(16, 32, 2, 2, 9)
    AQ3D_I8_CASE(32, 32, 2, 2, 4)
    AQ3D_I8_CASE(32, 64, 2, 2, 4)
    AQ3D_I8_CASE(64, 64, 2, 2, 2)
    AQ3D_I8_CASE(64, 128, 2, 2, 2)
    AQ3D_I8_CASE(128, 128, 2, 2, 1)
#undef AQ3D_I8_CASE
    return AQ3D_ERR_UNSUPPORTED;
}
