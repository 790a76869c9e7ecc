// Output-stationary gather-GEMM sparse convolution for gfx950 + the per-channel quantizer around it.
//
// What it replaces: QConvNd.forward (quant/quant.py:36-58) = TensorQuantizer(axis=1) on the features,
// per-out-channel fake-quant of the weight, then spconv's SubMConv3d/SparseConv3d gather-GEMM-scatter, followed by
// BatchNorm1d + ReLU (+ residual add) of pcdet/models/backbones_3d/spconv_backbone.py:8-27,51-67.
//
// Numerics: the reference multiplies fake-quantised fp32 values (integer code / scale) and accumulates in fp32.
// We reproduce exactly those operands — the activation scale is applied while the gather fills the LDS tile,
// the weight scale in the epilogue — and accumulate with the f32-input MFMA (v_mfma_f32_16x16x4_f32), which is
// bit-for-bit an fp32 fma chain. Only the summation order differs from the reference (≈1e-7 relative).
//
// Tiling (one 256-thread workgroup = 4 waves owns TM = 16*WR output rows x all COUT columns):
//   per kernel offset k: rows nbr[m][k] are gathered into As[TM][CIN] (zeros where there is no neighbour),
//   W_k^T is staged into Bs[COUT][CIN]; every wave then runs CIN/4 MFMA steps per 16x16 output block.
//   Within a 16-wide K chunk lane group g = lane>>4 supplies k = 4g + s at step s, so that each lane fetches its
//   A and B operands of four steps with ONE ds_read_b128 each (the k order inside an MFMA is free as long as
//   A and B agree).  Offsets for which no row of the tile has a neighbour are skipped (wave-uniform).
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SpconvArgs {
    const void *feats_in;
    const float *act_dequant;
    const void *weight;
    const float *w_dequant;
    const float *bias;
    const float *post_scale;
    const float *post_shift;
    const float *residual;
    const int *nbr;
    const int *count_out;
    float *feats_out;
    float *absmax_out;
    int cap_out, cin, kvol, relu;
    int cout_real;          // output channels that exist (the rest of the padded `cout` is zero weights); 0 = all
    // static PTQ (stem kernel only): the consumer's calibrated quantiser in this epilogue — codes i8 [cap_out][cout <= 32]
    const float *next_scale;
    int8_t *next_codes;
    float next_qmax;
};

// up to 8 independent layers of one shape in a single grid (blockIdx.y = problem): the VoxelNeXt head runs 72 small
// 1x1 sparse convs on one index level
#define AQ3D_SPCONV_MAX_GROUP 8
struct SpconvGroup {
    SpconvArgs p[AQ3D_SPCONV_MAX_GROUP];
};

template <int CIN_PAD, int COUT, int WR, int WC, bool A_I8, bool W_I8>
__global__ __launch_bounds__(256) void spconv_kernel(SpconvGroup grp) {
    const SpconvArgs a = grp.p[blockIdx.y];
    constexpr int TM = 16 * WR;
    constexpr int NT = COUT / (16 * WC);
    constexpr int LDA = CIN_PAD + 4;
    constexpr int C4 = CIN_PAD / 4;
    static_assert(WR * WC == 4, "4 waves");
    static_assert(COUT % (16 * WC) == 0, "COUT split");
    static_assert(CIN_PAD % 16 == 0, "CIN_PAD multiple of 16");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                                   // [TM][LDA]
    float *Bs = As + TM * LDA;                         // [COUT][LDA]
    int *nbr_s = reinterpret_cast<int *>(Bs + COUT * LDA);  // [TM][kvol]
    float *red = reinterpret_cast<float *>(nbr_s + TM * 27);  // [WR][COUT] abs-max partials

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WR, wc = wave / WR;
    const int g = lane >> 4, l15 = lane & 15;
    const int kvol = a.kvol, cin = a.cin;

    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const int ntiles = (n + TM - 1) / TM;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * TM;
        __syncthreads();  // previous tile's epilogue reads of nbr_s/red are done
        for (int t = tid; t < TM * kvol; t += 256) {
            int r = row0 + t / kvol;
            nbr_s[t] = (r < n) ? a.nbr[(size_t)row0 * kvol + t] : -1;
        }
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();

        for (int k = 0; k < kvol; ++k) {
            int has = 0;
            if (tid < TM) has = nbr_s[tid * kvol + k] >= 0;
            if (!__syncthreads_or(has)) continue;  // barrier: also fences the previous offset's LDS reads

            // ---- gather A rows (activation scale folded in here when the features are int8 codes)
            if constexpr (A_I8) {
                // 16 codes (16 B) per thread-item
                constexpr int C16 = CIN_PAD / 16;
                for (int t = tid; t < TM * C16; t += 256) {
                    int r = t / C16, c16 = t - r * C16;
                    int src = nbr_s[r * kvol + k];
                    float *dst = As + r * LDA + c16 * 16;
                    if (src >= 0) {
                        const int4 q = *reinterpret_cast<const int4 *>(static_cast<const int8_t *>(a.feats_in) +
                                                                        (size_t)src * cin + c16 * 16);
                        const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f32x4 v;
                            const float *dq = a.act_dequant + c16 * 16 + j * 4;
                            v.x = (float)(int8_t)(w[j] & 0xff) * dq[0];
                            v.y = (float)(int8_t)((w[j] >> 8) & 0xff) * dq[1];
                            v.z = (float)(int8_t)((w[j] >> 16) & 0xff) * dq[2];
                            v.w = (float)(int8_t)((w[j] >> 24) & 0xff) * dq[3];
                            *reinterpret_cast<f32x4 *>(dst + j * 4) = v;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(dst + j * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
            } else {
                const float *fin = static_cast<const float *>(a.feats_in);
                if ((cin & 3) == 0) {
                    for (int t = tid; t < TM * C4; t += 256) {
                        int r = t / C4, c4 = t - r * C4;
                        int src = nbr_s[r * kvol + k];
                        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (src >= 0 && c4 * 4 < cin) v = *reinterpret_cast<const f32x4 *>(fin + (size_t)src * cin + c4 * 4);
                        *reinterpret_cast<f32x4 *>(As + r * LDA + c4 * 4) = v;
                    }
                } else {
                    for (int t = tid; t < TM * CIN_PAD; t += 256) {
                        int r = t / CIN_PAD, c = t - r * CIN_PAD;
                        int src = nbr_s[r * kvol + k];
                        As[r * LDA + c] = (src >= 0 && c < cin) ? fin[(size_t)src * cin + c] : 0.f;
                    }
                }
            }
            // ---- stage W_k^T: global [kvol][COUT][cin] → Bs[COUT][LDA]
            if constexpr (W_I8) {
                constexpr int C16 = CIN_PAD / 16;
                const int8_t *wk = static_cast<const int8_t *>(a.weight) + (size_t)k * COUT * cin;
                for (int t = tid; t < COUT * C16; t += 256) {
                    int o = t / C16, c16 = t - o * C16;
                    const int4 q = *reinterpret_cast<const int4 *>(wk + (size_t)o * cin + c16 * 16);
                    const int w[4] = {q.x, q.y, q.z, q.w};
                    float *dst = Bs + o * LDA + c16 * 16;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v;
                        v.x = (float)(int8_t)(w[j] & 0xff);
                        v.y = (float)(int8_t)((w[j] >> 8) & 0xff);
                        v.z = (float)(int8_t)((w[j] >> 16) & 0xff);
                        v.w = (float)(int8_t)((w[j] >> 24) & 0xff);
                        *reinterpret_cast<f32x4 *>(dst + j * 4) = v;
                    }
                }
            } else {
                const float *wk = static_cast<const float *>(a.weight) + (size_t)k * COUT * cin;
                if ((cin & 3) == 0) {
                    for (int t = tid; t < COUT * C4; t += 256) {
                        int o = t / C4, c4 = t - o * C4;
                        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (c4 * 4 < cin) v = *reinterpret_cast<const f32x4 *>(wk + (size_t)o * cin + c4 * 4);
                        *reinterpret_cast<f32x4 *>(Bs + o * LDA + c4 * 4) = v;
                    }
                } else {
                    for (int t = tid; t < COUT * CIN_PAD; t += 256) {
                        int o = t / CIN_PAD, c = t - o * CIN_PAD;
                        Bs[o * LDA + c] = (c < cin) ? wk[(size_t)o * cin + c] : 0.f;
                    }
                }
            }
            __syncthreads();

            // ---- MFMA: 16 rows x (NT*16) cols per wave, K = CIN_PAD
            const float *arow = As + (wr * 16 + l15) * LDA + 4 * g;
            const float *brow = Bs + ((wc * NT) * 16 + l15) * LDA + 4 * g;
#pragma unroll 2
            for (int kc = 0; kc < CIN_PAD; kc += 16) {
                const f32x4 af = *reinterpret_cast<const f32x4 *>(arow + kc);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 bf = *reinterpret_cast<const f32x4 *>(brow + nt * 16 * LDA + kc);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, acc[nt], 0, 0, 0);
                }
            }
        }

        // ---- epilogue. D layout: col = lane&15 (out channel), row = 4*g + reg (tile row).
        float cmax[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int o = (wc * NT + nt) * 16 + l15;
            const float wdq = a.w_dequant ? a.w_dequant[o] : 1.f;
            const float b = a.bias ? a.bias[o] : 0.f;
            const float ps = a.post_scale ? a.post_scale[o] : 1.f;
            const float pt = a.post_shift ? a.post_shift[o] : 0.f;
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = row0 + wr * 16 + 4 * g + r;
                if (m < n) {
                    float v = acc[nt][r];
                    if (a.w_dequant) v = v * wdq;
                    if (a.bias) v = v + b;
                    if (a.post_scale) v = v * ps + pt;
                    if (a.residual) v += a.residual[(size_t)m * COUT + o];
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
                // monotone: a stale read only costs an extra atomic
                if (mx > a.absmax_out[o]) aq3d_atomic_max_pos(&a.absmax_out[o], mx);
            }
        }
    }
}

// fp32 activations x int8 weight codes on the fp16 matrix pipe (VoxelNeXt's weight-only QConv2d head: grouped 3x3 / 1x1 sparse
// 2-D convs). Same tiling, gather and epilogue as spconv_kernel<.., false, true>; the staged operands are fp16 instead of fp32:
// an activation x goes to LDS as hi = f16(x) and lo' = f16((x - hi) * 2^11) (x == hi + lo exactly; 22 significant bits), a weight
// code is an exact small integer in fp16, and every 32-deep K chunk costs two v_mfma_f32_16x16x32_f16 (hi into one fp32
// accumulator, lo' into a second one that the epilogue folds in as acc + 2^-11 * acc_lo) instead of eight fp32 MFMAs of twice
// the latency. Products are exact in fp32; only the 2^-22 tail of each activation is dropped. (Rounds 1-2 scaled lo' by 2^-11 IN
// fp16 and used one accumulator: for |x| below ~2^-3 that product is an fp16 subnormal and the bits the scaling was meant to keep
// were lost — ADVICE r2; tests/test_sparse_gpu.py::test_spconv_f32_small_and_large_magnitudes.)
typedef _Float16 sp_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sp_f16x4 __attribute__((ext_vector_type(4)));
// W32: fp32 weights (an un-quantised layer), split into fp16 hi | lo planes while staged, three MFMAs per product like
// conv2d_f16s_kernel (a_hi*w_hi + a_lo*w_hi + a_hi*w_lo; the lo*lo term, ~2^-22 relative, is dropped). The fp32 matrix pipe
// (spconv_kernel) took 242 us for VoxelNeXt's one 128 -> 128 fp32 layer.
// WSPLIT (with W32): the fp32 weights arrive already split, f16 [kvol][Cout][2][Cin] (aq3d_conv2d_split_weight; w_mode AQ3D_W_F16S) — the
// split is the same for every row tile and kernel offset, and a 128 x 128 fp32 tile split by 256 threads per offset was most of the
// un-quantised 128 -> 128 layer's 180 us.
// (Round 4 measured the int8 codes pre-expanded to fp16 — the weight tile of an offset as a plain 16-byte copy into LDS instead of 64
// conversions per thread: 95.5 against 92 us per grouped launch of VoxelNeXt's weight-only 128 -> 128 convs, i.e. nothing; removed.)
template <int CIN, int COUT, int WR, int WC, bool W32 = false, bool WSPLIT = false>
__global__ __launch_bounds__(256) void spconv_f16w_kernel(SpconvGroup grp) {
    const SpconvArgs a = grp.p[blockIdx.y];
    constexpr int TM = 16 * WR;
    constexpr int NT = COUT / (16 * WC);
    constexpr int LDH = CIN + 8;            // halves per LDS row (16-byte pad)
    constexpr int C4 = CIN / 4, C16 = CIN / 16;
    static_assert(WR * WC == 4, "4 waves");
    static_assert(COUT % (16 * WC) == 0 && CIN % 32 == 0, "shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    _Float16 *Ah = reinterpret_cast<_Float16 *>(lds_raw);          // [TM][LDH]
    _Float16 *Al = Ah + TM * LDH;                                   // [TM][LDH]
    _Float16 *Bh = Al + TM * LDH;                                   // [COUT][LDH]
    _Float16 *Bl = Bh + COUT * LDH;                                 // [COUT][LDH]  (W32 only)
    int *nbr_s = reinterpret_cast<int *>(Bh + (W32 ? 2 : 1) * COUT * LDH);          // [TM][kvol]
    float *red = reinterpret_cast<float *>(nbr_s + TM * 27);        // [WR][COUT] abs-max partials

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WR, wc = wave / WR;
    const int g = lane >> 4, l15 = lane & 15;
    const int kvol = a.kvol;

    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const int ntiles = (n + TM - 1) / TM;
    const float *fin = static_cast<const float *>(a.feats_in);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * TM;
        __syncthreads();  // previous tile's epilogue reads of nbr_s/red are done
        for (int t = tid; t < TM * kvol; t += 256) {
            int r = row0 + t / kvol;
            nbr_s[t] = (r < n) ? a.nbr[(size_t)row0 * kvol + t] : -1;
        }
        f32x4 acc[NT], accl[NT];        // products with the hi halves | with one lo' half (scaled by 2^11)
#pragma unroll
        for (int i = 0; i < NT; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; accl[i] = acc[i]; }
        __syncthreads();

        for (int k = 0; k < kvol; ++k) {
            int has = 0;
            if (tid < TM) has = nbr_s[tid * kvol + k] >= 0;
            if (!__syncthreads_or(has)) continue;  // barrier: also fences the previous offset's LDS reads
            for (int t = tid; t < TM * C4; t += 256) {
                const int r = t / C4, c4 = t - r * C4;
                const int src = nbr_s[r * kvol + k];
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (src >= 0) v = *reinterpret_cast<const f32x4 *>(fin + (size_t)src * CIN + c4 * 4);
                sp_f16x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h[j] = (_Float16)v[j];
                    l[j] = (_Float16)__fmul_rn(__fsub_rn(v[j], (float)h[j]), 2048.f);
                }
                *reinterpret_cast<sp_f16x4 *>(Ah + r * LDH + c4 * 4) = h;
                *reinterpret_cast<sp_f16x4 *>(Al + r * LDH + c4 * 4) = l;
            }
            if (W32 && WSPLIT) {
                constexpr int C8 = CIN / 8;
                const _Float16 *wk16 = static_cast<const _Float16 *>(a.weight) + (size_t)k * COUT * 2 * CIN;
                for (int t = tid; t < COUT * C8; t += 256) {
                    const int o = t / C8, c8 = t - o * C8;
                    *reinterpret_cast<sp_f16x8 *>(Bh + o * LDH + c8 * 8) = *reinterpret_cast<const sp_f16x8 *>(wk16 + (size_t)(o * 2) * CIN + c8 * 8);
                    *reinterpret_cast<sp_f16x8 *>(Bl + o * LDH + c8 * 8) = *reinterpret_cast<const sp_f16x8 *>(wk16 + (size_t)(o * 2 + 1) * CIN + c8 * 8);
                }
            }
            if (W32 && !WSPLIT) {
                const float *wk32 = static_cast<const float *>(a.weight) + (size_t)k * COUT * CIN;
                for (int t = tid; t < COUT * C4; t += 256) {
                    const int o = t / C4, c4 = t - o * C4;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(wk32 + (size_t)o * CIN + c4 * 4);
                    sp_f16x4 h, l;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        h[j] = (_Float16)v[j];
                        l[j] = (_Float16)__fmul_rn(__fsub_rn(v[j], (float)h[j]), 2048.f);
                    }
                    *reinterpret_cast<sp_f16x4 *>(Bh + o * LDH + c4 * 4) = h;
                    *reinterpret_cast<sp_f16x4 *>(Bl + o * LDH + c4 * 4) = l;
                }
            }
            const int8_t *wk = static_cast<const int8_t *>(a.weight) + (size_t)k * COUT * CIN;
            for (int t = tid; !W32 && t < COUT * C16; t += 256) {
                const int o = t / C16, c16 = t - o * C16;
                const int4 q = *reinterpret_cast<const int4 *>(wk + (size_t)o * CIN + c16 * 16);
                const int w[4] = {q.x, q.y, q.z, q.w};
                sp_f16x8 lo8, hi8;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    lo8[j * 4 + 0] = (_Float16)(float)(int8_t)(w[j] & 0xff);
                    lo8[j * 4 + 1] = (_Float16)(float)(int8_t)((w[j] >> 8) & 0xff);
                    lo8[j * 4 + 2] = (_Float16)(float)(int8_t)((w[j] >> 16) & 0xff);
                    lo8[j * 4 + 3] = (_Float16)(float)(int8_t)((w[j] >> 24) & 0xff);
                    hi8[j * 4 + 0] = (_Float16)(float)(int8_t)(w[2 + j] & 0xff);
                    hi8[j * 4 + 1] = (_Float16)(float)(int8_t)((w[2 + j] >> 8) & 0xff);
                    hi8[j * 4 + 2] = (_Float16)(float)(int8_t)((w[2 + j] >> 16) & 0xff);
                    hi8[j * 4 + 3] = (_Float16)(float)(int8_t)((w[2 + j] >> 24) & 0xff);
                }
                *reinterpret_cast<sp_f16x8 *>(Bh + o * LDH + c16 * 16) = lo8;
                *reinterpret_cast<sp_f16x8 *>(Bh + o * LDH + c16 * 16 + 8) = hi8;
            }
            __syncthreads();

            const _Float16 *ah = Ah + (wr * 16 + l15) * LDH + 8 * g;
            const _Float16 *al = Al + (wr * 16 + l15) * LDH + 8 * g;
            const _Float16 *bh = Bh + ((wc * NT) * 16 + l15) * LDH + 8 * g;
#pragma unroll 2
            for (int kc = 0; kc < CIN; kc += 32) {
                const sp_f16x8 fh = *reinterpret_cast<const sp_f16x8 *>(ah + kc);
                const sp_f16x8 fl = *reinterpret_cast<const sp_f16x8 *>(al + kc);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const sp_f16x8 fb = *reinterpret_cast<const sp_f16x8 *>(bh + nt * 16 * LDH + kc);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, fb, acc[nt], 0, 0, 0);
                    accl[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, fb, accl[nt], 0, 0, 0);
                    if (W32) {
                        const sp_f16x8 fbl = *reinterpret_cast<const sp_f16x8 *>(bh + COUT * LDH + nt * 16 * LDH + kc);
                        accl[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, fbl, accl[nt], 0, 0, 0);
                    }
                }
            }
        }

        // ---- epilogue (identical to spconv_kernel). D layout: col = lane&15 (out channel), row = 4*g + reg (tile row).
        float cmax[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int o = (wc * NT + nt) * 16 + l15;
            const float wdq = a.w_dequant ? a.w_dequant[o] : 1.f;
            const float b = a.bias ? a.bias[o] : 0.f;
            const float ps = a.post_scale ? a.post_scale[o] : 1.f;
            const float pt = a.post_shift ? a.post_shift[o] : 0.f;
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = row0 + wr * 16 + 4 * g + r;
                if (m < n) {
                    float v = __fadd_rn(acc[nt][r], __fmul_rn(accl[nt][r], 1.0f / 2048.f));
                    if (a.w_dequant) v = v * wdq;
                    if (a.bias) v = v + b;
                    if (a.post_scale) v = v * ps + pt;
                    if (a.residual) v += a.residual[(size_t)m * COUT + o];
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

template <int CIN_PAD, int COUT, int WR, int WC>
static int spconv_launch_modes(const SpconvGroup &grp, int n, int act_mode, int w_mode, hipStream_t stream) {
    const SpconvArgs &a = grp.p[0];
    constexpr int TM = 16 * WR;
    constexpr size_t lds = (size_t)(TM + COUT) * (CIN_PAD + 4) * 4 + (size_t)TM * 27 * 4 + (size_t)WR * COUT * 4;
    const int tiles = aq3d_div_up(a.cap_out, TM);
    const int grid = tiles < 2048 ? (tiles < 1 ? 1 : tiles) : 2048;
#define AQ3D_SPCONV_GO(AI8, WI8)                                                                               \
    do {                                                                                                       \
        auto kern = spconv_kernel<CIN_PAD, COUT, WR, WC, AI8, WI8>;                                            \
        if (lds > 64 * 1024)                                                                                   \
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, dim3(grid, n), dim3(256), lds, stream, grp);                                   \
    } while (0)
    const bool f16w = true;
    if constexpr (CIN_PAD % 32 == 0) {
        if (f16w && act_mode == AQ3D_ACT_F32 && w_mode == AQ3D_W_I8 && a.cin == CIN_PAD) {
            bool same = true;
            for (int i = 1; i < n; ++i) same = same && grp.p[i].cin == CIN_PAD;
            if constexpr (COUT == 128 && CIN_PAD == 128) if (same) {     // 64-row tiles (4 x 1 waves): the weights are re-streamed per tile (125 -> 100 us)
                constexpr int TM2 = 64;
                constexpr size_t lds3 = (size_t)(2 * TM2 + COUT) * (CIN_PAD + 8) * 2 + (size_t)TM2 * 27 * 4 + (size_t)4 * COUT * 4;
                const int tiles2 = aq3d_div_up(a.cap_out, TM2);
                const int grid2 = tiles2 < 2048 ? (tiles2 < 1 ? 1 : tiles2) : 2048;
                auto kern = spconv_f16w_kernel<CIN_PAD, COUT, 4, 1>;
                if (lds3 > 64 * 1024)
                    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
                hipLaunchKernelGGL(kern, dim3(grid2, n), dim3(256), lds3, stream, grp);
                AQ3D_CHECK_LAUNCH("aq3d_spconv");
                return AQ3D_OK;
            }

            if (same) {
                constexpr size_t lds2 = (size_t)(2 * TM + COUT) * (CIN_PAD + 8) * 2 + (size_t)TM * 27 * 4 + (size_t)WR * COUT * 4;
                auto kern = spconv_f16w_kernel<CIN_PAD, COUT, WR, WC>;
                if (lds2 > 64 * 1024)
                    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                hipLaunchKernelGGL(kern, dim3(grid, n), dim3(256), lds2, stream, grp);
                AQ3D_CHECK_LAUNCH("aq3d_spconv");
                return AQ3D_OK;
            }
        }
    }
    if constexpr (CIN_PAD == 128 && (COUT == 128 || COUT == 16)) {
        if (act_mode == AQ3D_ACT_F32 && (w_mode == AQ3D_W_F32 || w_mode == AQ3D_W_F16S) && a.cin == CIN_PAD) {      // un-quantised 128 -> 128 / 128 -> 16 layers: fp16 pipe, split operands
            bool same = true;
            for (int i = 1; i < n; ++i) same = same && grp.p[i].cin == CIN_PAD;
            if (same) {
                constexpr int TM2 = 64;
                constexpr size_t lds3 = (size_t)(2 * TM2 + 2 * COUT) * (CIN_PAD + 8) * 2 + (size_t)TM2 * 27 * 4 + (size_t)4 * COUT * 4;
                const int tiles2 = aq3d_div_up(a.cap_out, TM2);
                const int grid2 = tiles2 < 2048 ? (tiles2 < 1 ? 1 : tiles2) : 2048;
                if (w_mode == AQ3D_W_F16S) {
                    auto kern = spconv_f16w_kernel<CIN_PAD, COUT, 4, 1, true, true>;
                    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
                    hipLaunchKernelGGL(kern, dim3(grid2, n), dim3(256), lds3, stream, grp);
                } else {
                    auto kern = spconv_f16w_kernel<CIN_PAD, COUT, 4, 1, true>;
                    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
                    hipLaunchKernelGGL(kern, dim3(grid2, n), dim3(256), lds3, stream, grp);
                }
                AQ3D_CHECK_LAUNCH("aq3d_spconv");
                return AQ3D_OK;
            }
        }
    }
    if (act_mode == AQ3D_ACT_I8 && w_mode == AQ3D_W_I8) AQ3D_SPCONV_GO(true, true);
    else if (act_mode == AQ3D_ACT_I8) AQ3D_SPCONV_GO(true, false);
    else if (w_mode == AQ3D_W_I8) AQ3D_SPCONV_GO(false, true);
    else AQ3D_SPCONV_GO(false, false);
#undef AQ3D_SPCONV_GO
    AQ3D_CHECK_LAUNCH("aq3d_spconv");
    return AQ3D_OK;
}

// ---------------------------------------------------------------- fp32 sparse conv, 128 -> 1..4 channels: plain VALU
// VoxelNeXt's head output convs (SubMConv2d 3x3, 128 -> 1..3 channels, un-quantised: voxelnext_head.py:13-47) were padded to a
// 16-column MFMA tile: 30 of them per frame at ~10 us each, all gather and operand staging. Here half a wave owns one output row:
// lane l of the half holds channels 4l .. 4l+3 of the row's neighbours (one coalesced 512-byte read per kernel offset) and keeps the
// KV x CO x 4 weights of its channels in registers for the whole launch; the CO sums are fp32 FMA chains (offset, then channel) folded
// over the 32 lanes by a butterfly. Fixed order, fp32 throughout.
template <int CO, int KV>
__device__ __forceinline__ void spconv_small_body(const SpconvArgs &a, int cout_ld) {
    constexpr int CIN = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l = lane & 31;
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const float *x = static_cast<const float *>(a.feats_in);
    const float *w = static_cast<const float *>(a.weight);            // [KV][cout_ld][CIN]
    f32x4 wr[KV][CO];
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int o = 0; o < CO; ++o) wr[k][o] = *reinterpret_cast<const f32x4 *>(w + ((size_t)k * cout_ld + o) * CIN + 4 * l);
    float bs[CO], ps[CO], pt[CO], mx[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        bs[o] = a.bias ? a.bias[o] : 0.f;
        ps[o] = a.post_scale ? a.post_scale[o] : 1.f;
        pt[o] = a.post_scale ? a.post_shift[o] : 0.f;
        mx[o] = 0.f;
    }
    // U row pairs per round (1x1 convs: four — a round is a chain of dependent latencies: gather-table entry, row, butterfly, store)
    constexpr int U = KV == 1 ? 4 : 1;
    const int nw = gridDim.x * (blockDim.x >> 6);
    for (int r0 = (blockIdx.x * (blockDim.x >> 6) + wave) * 2 * U; r0 < n; r0 += nw * 2 * U) {
        int src[U][KV];
        f32x4 xv[U][KV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + 2 * u + half;
#pragma unroll
            for (int k = 0; k < KV; ++k) src[u][k] = r < n ? a.nbr[(size_t)r * KV + k] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                xv[u][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (src[u][k] >= 0) xv[u][k] = *reinterpret_cast<const f32x4 *>(x + (size_t)src[u][k] * CIN + 4 * l);
            }
        float acc[U][CO];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < KV; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t = __fmaf_rn(xv[u][k][j], wr[k][o][j], t);
                acc[u][o] = t;
            }
#pragma unroll
        for (int sft = 1; sft < 32; sft <<= 1)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int o = 0; o < CO; ++o) acc[u][o] = __fadd_rn(acc[u][o], __shfl_xor(acc[u][o], sft, 64));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + 2 * u + half;
            if (l == 0 && r < n) {
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    float v = acc[u][o];
                    if (a.bias) v = __fadd_rn(v, bs[o]);
                    if (a.post_scale) v = __fadd_rn(__fmul_rn(v, ps[o]), pt[o]);
                    if (a.relu) v = fmaxf(v, 0.f);
                    a.feats_out[(size_t)r * cout_ld + o] = v;
                    mx[o] = fmaxf(mx[o], fabsf(v));
                }
            }
        }
    }
    if (a.absmax_out) {         // (uniform per workgroup) one atomic per workgroup and channel, and only when it would raise the value
        __shared__ float red[4][4];
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            const float m = fmaxf(mx[o], __shfl_xor(mx[o], 32, 64));       // lanes 0 and 32 hold the two halves' maxima
            if (lane == 0) red[wave & 3][o] = m;
        }
        __syncthreads();
        if (threadIdx.x < CO) {
            const int o = threadIdx.x;
            float m = red[0][o];
            for (int w_ = 1; w_ < (int)(blockDim.x >> 6) && w_ < 4; ++w_) m = fmaxf(m, red[w_][o]);
            if (m > 0.f && m > a.absmax_out[o]) aq3d_atomic_max_pos(&a.absmax_out[o], m);
        }
    }
}
template <int KV>
__global__ __launch_bounds__(256) void spconv_small_kernel(SpconvGroup grp, int cout_ld) {
    const SpconvArgs a = grp.p[blockIdx.y];
    switch (a.cout_real) {      // uniform per workgroup
    case 1: spconv_small_body<1, KV>(a, cout_ld); break;
    case 2: spconv_small_body<2, KV>(a, cout_ld); break;
    case 3: spconv_small_body<3, KV>(a, cout_ld); break;
    default: spconv_small_body<4, KV>(a, cout_ld); break;
    }
}

// ---------------------------------------------------------------- stem layer (conv_input: Cin <= 8 raw point features -> 16)
// K = 27 * Cin is ~135: far too thin for tiles, LDS staging and barriers (the generic kernel takes 42 us for 30 k rows, all of
// it latency). Plain VALU instead: thread = one output row x 4 output channels (blockIdx.y = channel quad, so the 4*Cin
// weights of a kernel offset are wave-uniform scalar operands), fp32 fma chain in (offset, channel) order — the same chain
// as the MFMA kernel and the oracle's per-offset matmul. (16 channels per thread — every input gathered once instead of once per
// channel quad — measured slower, 46 against 37 us: the kernel is a latency chain and the four quads are its parallelism.)
// Round 4 first tried to turn the six dependent memory rounds into two while still multiplying all 27 offsets (gather table through
// LDS as one coalesced block, all 27 x Cin feature loads in flight, eight channels per thread): 57 us with branches around the loads,
// 57 us with bounded loads, 224 us with the weights in LDS (512 VGPRs + 2.9 KB of scratch) against this kernel's 34 us — time followed
// the per-thread length of the (scalar weight load -> fma) chain, 540 operands here, 1080 there. What worked is further down
// (spconv_stem_pairs_kernel): not multiplying the ~24 of 27 offsets a voxel does not have. This kernel stays for other kernel volumes
// and channel counts.
template <int CIN>
__global__ __launch_bounds__(256) void spconv_stem_kernel(SpconvArgs a, int cout) {
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int o0 = blockIdx.y * 4;
    const float *x = static_cast<const float *>(a.feats_in);
    const float *w = static_cast<const float *>(a.weight);        // [kvol][cout][CIN]
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // chunks of 9 kernel offsets: the 9 gather-table entries, then the 9 x CIN feature values, are all in flight together
    // (bounded buffer loads: a missing neighbour / a row past the end is an out-of-range offset -> zeros, no branches); a
    // zero activation adds +0 to the chain, so the sum equals the one over the present neighbours only
    const auto rsrc_n = __builtin_amdgcn_make_buffer_rsrc((void *)a.nbr, 0, n * a.kvol * 4, 0x00020000);
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, 0x7ffffff0, 0x00020000);
    for (int k0 = 0; k0 < a.kvol; k0 += 9) {
        int src[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const unsigned off = (row < n && k0 + u < a.kvol) ? (unsigned)((row * a.kvol + k0 + u) * 4) : 0xfffffff0u;
            src[u] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc_n, off, 0, 0) - ((row < n && k0 + u < a.kvol) ? 0 : 1);
        }
        float xv[9][CIN];
#pragma unroll
        for (int u = 0; u < 9; ++u)
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const unsigned off = src[u] >= 0 ? (unsigned)((src[u] * CIN + c) * 4) : 0xfffffff0u;
                xv[u][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, off, 0, 0));
            }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            if (k0 + u >= a.kvol) break;
            const float *wk = w + ((size_t)(k0 + u) * cout + o0) * CIN;    // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = 0.f;
#pragma unroll
                for (int c = 0; c < CIN; ++c) t = __fmaf_rn(xv[u][c], wk[j * CIN + c], t);
                acc[j] = __fadd_rn(acc[j], t);
            }
        }
    }
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = o0 + j;
            float v = acc[j];
            if (a.bias) v = __fadd_rn(v, a.bias[o]);
            if (a.post_scale) v = __fadd_rn(__fmul_rn(v, a.post_scale[o]), a.post_shift[o]);
            if (a.residual) v = __fadd_rn(v, a.residual[(size_t)row * cout + o]);
            if (a.relu) v = fmaxf(v, 0.f);
            a.feats_out[(size_t)row * cout + o] = v;
            if (a.next_codes) {       // cout < 64: identity code layout
                float q = rintf(__fmul_rn(v, a.next_scale[o]));
                q = fminf(fmaxf(q, -a.next_qmax - 1.f), a.next_qmax);
                a.next_codes[(size_t)row * cout + o] = (int8_t)(int)q;
            }
            mx[j] = fabsf(v);
        }
    }
    if (a.absmax_out) {
        __shared__ float red[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m = aq3d_wave_max(mx[j]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = m;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            const float m = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
            if (m > a.absmax_out[o0 + threadIdx.x]) aq3d_atomic_max_pos(&a.absmax_out[o0 + threadIdx.x], m);
        }
    }
}

// Round 4, the stem for the 3 x 3 x 3 / 16-output-channel case: work that follows the (row, offset) PAIRS that exist. A LiDAR voxel
// has ~3 of its 27 neighbours; the kernel above multiplies all 27 (zeros for the missing ones) and gathers every input four times
// (once per channel quad), and its time follows the per-thread chain of 540 scalar-operand fmas. Here a thread owns one output row
// and ALL 16 output channels: the 256 x 27 table entries of the workgroup arrive as one coalesced block in LDS, every thread compacts
// its own 27 entries in place to the (offset, source row) pairs it has, then loops over those — one 5-float gather, 80 fmas with the
// weights of that offset read from LDS — in ascending offset order. The sums are the chains of the kernel above with the +0 terms of
// the missing offsets left out: identical bits.
template <int CIN>
__global__ __launch_bounds__(256) void spconv_stem_pairs_kernel(SpconvArgs a) {
    // (measured, rocprof averages at 30 000 rows: 256 rows per workgroup and one pair per round 15.6 us; 128 rows 16.6; 2 / 4 pairs'
    // gathers in flight per round 17.6 / 19.6 — the gathers are not what the loop waits for)
    constexpr int KV = 27, CO = 16, WS = 20, NT = 256, G = 1;   // WS: floats per (offset, channel) weight row in LDS (16 + 4 pad); G: pairs in flight
    __shared__ int s_nbr[NT * KV];
    __shared__ __attribute__((aligned(16))) float s_w[KV * CIN * WS];
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const int row0 = blockIdx.x * NT;
    if (row0 >= n) return;                                   // (uniform) buffers are sized for the capacity, not the frame
    const int row = row0 + threadIdx.x;
    const float *x = static_cast<const float *>(a.feats_in);
    const float *w = static_cast<const float *>(a.weight);        // [kvol][16][CIN]
    {
        const int valid = (n - row0 < NT ? n - row0 : NT) * KV;
        const int *src = a.nbr + (size_t)row0 * KV;
#pragma unroll
        for (int i = 0; i < KV; ++i) {
            const int e = i * NT + threadIdx.x;
            s_nbr[e] = e < valid ? src[e] : -1;
        }
    }
    for (int i = threadIdx.x; i < KV * CIN * CO; i += NT) {        // s_w[(u * CIN + c) * WS + o] <- w[u][o][c]
        const int o = i % CO, c = (i / CO) % CIN, u = i / (CO * CIN);
        s_w[(u * CIN + c) * WS + o] = w[((size_t)u * CO + o) * CIN + c];
    }
    __syncthreads();
    // in-place compaction of the thread's own 27 entries: (offset << 26) | source row for the neighbours that exist
    int *mine = s_nbr + threadIdx.x * KV;                         // stride 27 words: conflict-free
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < KV; ++u) {
        const int sr = mine[u];
        if (sr >= 0) mine[cnt++] = (u << 26) | sr;
    }
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, 0x7ffffff0, 0x00020000);
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    // G pairs per round: their G x CIN gathers are in flight together (bounded loads: a slot past the thread's last pair is an
    // out-of-range offset = zeros and adds +0 to the chain), then G x 16 x CIN fmas in ascending offset order
    for (int j0 = 0; j0 < cnt; j0 += G) {
        int u[G];
        float xv[G][CIN];
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const bool live = j0 + q < cnt;
            const int p = live ? mine[j0 + q] : 0;
            u[q] = (int)((unsigned)p >> 26);
            const int sr = p & 0x03ffffff;
#pragma unroll
            for (int c = 0; c < CIN; ++c)
                xv[q][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, live ? (unsigned)((sr * CIN + c) * 4) : 0xfffffff0u, 0, 0));
        }
#pragma unroll
        for (int q = 0; q < G; ++q) {
            float t[CO];
#pragma unroll
            for (int o = 0; o < CO; ++o) t[o] = 0.f;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float *wr = s_w + (u[q] * CIN + c) * WS;
#pragma unroll
                for (int o4 = 0; o4 < CO; o4 += 4) {
                    const float4 wv = *reinterpret_cast<const float4 *>(wr + o4);
                    t[o4 + 0] = __fmaf_rn(xv[q][c], wv.x, t[o4 + 0]);
                    t[o4 + 1] = __fmaf_rn(xv[q][c], wv.y, t[o4 + 1]);
                    t[o4 + 2] = __fmaf_rn(xv[q][c], wv.z, t[o4 + 2]);
                    t[o4 + 3] = __fmaf_rn(xv[q][c], wv.w, t[o4 + 3]);
                }
            }
#pragma unroll
            for (int o = 0; o < CO; ++o) acc[o] = __fadd_rn(acc[o], t[o]);
        }
    }
    float mx[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) mx[o] = 0.f;
    if (row < n) {
        float out[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            float v = acc[o];
            if (a.bias) v = __fadd_rn(v, a.bias[o]);
            if (a.post_scale) v = __fadd_rn(__fmul_rn(v, a.post_scale[o]), a.post_shift[o]);
            if (a.residual) v = __fadd_rn(v, a.residual[(size_t)row * CO + o]);
            if (a.relu) v = fmaxf(v, 0.f);
            out[o] = v;
            if (a.next_codes) {       // static PTQ: the consumer's quantiser (identity code layout)
                float q = rintf(__fmul_rn(v, a.next_scale[o]));
                q = fminf(fmaxf(q, -a.next_qmax - 1.f), a.next_qmax);
                a.next_codes[(size_t)row * CO + o] = (int8_t)(int)q;
            }
            mx[o] = fabsf(v);
        }
        float4 *dst = reinterpret_cast<float4 *>(a.feats_out + (size_t)row * CO);
#pragma unroll
        for (int o = 0; o < CO; o += 4) dst[o / 4] = make_float4(out[o], out[o + 1], out[o + 2], out[o + 3]);
    }
    if (a.absmax_out) {
        __shared__ float red[NT / 64][CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            const float m = aq3d_wave_max(mx[o]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][o] = m;
        }
        __syncthreads();
        if (threadIdx.x < CO) {
            float m = red[0][threadIdx.x];
#pragma unroll
            for (int q = 1; q < NT / 64; ++q) m = fmaxf(m, red[q][threadIdx.x]);
            if (m > a.absmax_out[threadIdx.x]) aq3d_atomic_max_pos(&a.absmax_out[threadIdx.x], m);
        }
    }
}

static int spconv_group_impl(const aq3d_spconv_problem *probs, int32_t n, int32_t act_mode, int32_t w_mode, int32_t cap_out,
                             int32_t cap_in, int32_t cin, int32_t cout, int32_t kvol, int32_t flags, const float *next_scale,
                             int8_t *next_codes, float next_qmax, void *stream_);

extern "C" int aq3d_spconv(const void *feats_in, int32_t act_mode, const float *act_dequant, const void *weight,
                           int32_t w_mode, const float *w_dequant, const float *bias, const float *post_scale,
                           const float *post_shift, const float *residual, const int32_t *nbr,
                           const int32_t *count_out, int32_t cap_out, int32_t cap_in, int32_t cin, int32_t cout,
                           int32_t kvol, int32_t flags, float *feats_out, float *absmax_out, const float *next_scale,
                           int8_t *next_codes, int32_t next_bits, void *stream_) {
    AQ3D_REQUIRE(feats_in && weight && nbr && count_out && feats_out, "aq3d_spconv: null argument");
    AQ3D_REQUIRE((next_scale == nullptr) == (next_codes == nullptr), "aq3d_spconv: next_scale / next_codes must come together");
    if (next_codes) {       // only the stem kernel carries the consumer's static quantiser
        AQ3D_REQUIRE(act_mode == AQ3D_ACT_F32 && w_mode == AQ3D_W_F32 && cin <= 8 && (cout & 3) == 0 && cout <= 32 && next_bits >= 2 && next_bits <= 8,
                     "aq3d_spconv: next_codes is supported by the stem layer only (fp32, cin <= 8, cout <= 32)");
    }
    AQ3D_REQUIRE(cap_out > 0 && kvol >= 1 && kvol <= 27, "aq3d_spconv: cap_out=%d kvol=%d", cap_out, kvol);
    AQ3D_REQUIRE(cin >= 1 && cin <= 128, "aq3d_spconv: cin=%d not in 1..128", cin);
    AQ3D_REQUIRE(act_mode != AQ3D_ACT_I8 || ((cin % 16) == 0 && act_dequant), "aq3d_spconv: int8 activations need cin%%16==0 and act_dequant");
    AQ3D_REQUIRE(w_mode != AQ3D_W_I8 || ((cin % 16) == 0 && w_dequant), "aq3d_spconv: int8 weights need cin%%16==0 and w_dequant");
    AQ3D_REQUIRE((post_scale == nullptr) == (post_shift == nullptr), "aq3d_spconv: post_scale/post_shift must come together");
    aq3d_spconv_problem q;
    q.feats_in = feats_in; q.act_dequant = act_dequant; q.weight = weight; q.w_dequant = w_dequant; q.bias = bias;
    q.post_scale = post_scale; q.post_shift = post_shift; q.residual = residual; q.nbr = nbr; q.count_out = count_out;
    q.feats_out = feats_out; q.absmax_out = absmax_out; q.cout_real = 0;
    return spconv_group_impl(&q, 1, act_mode, w_mode, cap_out, cap_in, cin, cout, kvol, flags, next_scale, next_codes,
                             next_codes ? (float)((1 << (next_bits - 1)) - 1) : 0.f, stream_);
}

extern "C" int aq3d_spconv_group(const aq3d_spconv_problem *probs, int32_t n, int32_t act_mode, int32_t w_mode, int32_t cap_out,
                                 int32_t cap_in, int32_t cin, int32_t cout, int32_t kvol, int32_t flags, void *stream_) {
    return spconv_group_impl(probs, n, act_mode, w_mode, cap_out, cap_in, cin, cout, kvol, flags, nullptr, nullptr, 0.f, stream_);
}

static int spconv_group_impl(const aq3d_spconv_problem *probs, int32_t n, int32_t act_mode, int32_t w_mode, int32_t cap_out,
                             int32_t cap_in, int32_t cin, int32_t cout, int32_t kvol, int32_t flags, const float *next_scale,
                             int8_t *next_codes, float next_qmax, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    (void)cap_in;
    AQ3D_REQUIRE(probs && n >= 1 && n <= AQ3D_SPCONV_MAX_GROUP, "aq3d_spconv_group: n=%d not in 1..%d", n, AQ3D_SPCONV_MAX_GROUP);
    AQ3D_REQUIRE(cap_out > 0 && kvol >= 1 && kvol <= 27, "aq3d_spconv: cap_out=%d kvol=%d", cap_out, kvol);
    AQ3D_REQUIRE(cin >= 1 && cin <= 128, "aq3d_spconv: cin=%d not in 1..128", cin);
    SpconvGroup grp;
    for (int i = 0; i < n; ++i) {
        const aq3d_spconv_problem &q = probs[i];
        AQ3D_REQUIRE(q.feats_in && q.weight && q.nbr && q.count_out && q.feats_out, "aq3d_spconv: null argument");
        AQ3D_REQUIRE(act_mode != AQ3D_ACT_I8 || ((cin % 16) == 0 && q.act_dequant), "aq3d_spconv: int8 activations need cin%%16==0 and act_dequant");
        AQ3D_REQUIRE(w_mode != AQ3D_W_I8 || ((cin % 16) == 0 && q.w_dequant), "aq3d_spconv: int8 weights need cin%%16==0 and w_dequant");
        AQ3D_REQUIRE((q.post_scale == nullptr) == (q.post_shift == nullptr), "aq3d_spconv: post_scale/post_shift must come together");
        AQ3D_REQUIRE(w_mode != AQ3D_W_F16S || (act_mode == AQ3D_ACT_F32 && cin == 128 && (cout == 128 || cout == 16)),
                     "aq3d_spconv: pre-split fp16 weight planes are taken for f32 activations, cin = 128, cout = 128 | 16 only");
        SpconvArgs &a = grp.p[i];
        a.feats_in = q.feats_in; a.act_dequant = q.act_dequant; a.weight = q.weight; a.w_dequant = q.w_dequant;
        a.bias = q.bias; a.post_scale = q.post_scale; a.post_shift = q.post_shift; a.residual = q.residual;
        a.nbr = q.nbr; a.count_out = q.count_out; a.feats_out = q.feats_out; a.absmax_out = q.absmax_out;
        a.cap_out = cap_out; a.cin = cin; a.kvol = kvol; a.relu = (flags & AQ3D_FLAG_RELU) ? 1 : 0;
        a.cout_real = q.cout_real > 0 && q.cout_real < cout ? q.cout_real : cout;
        a.next_scale = next_scale; a.next_codes = next_codes; a.next_qmax = next_qmax;
    }
    for (int i = n; i < AQ3D_SPCONV_MAX_GROUP; ++i) grp.p[i] = grp.p[0];
    const SpconvArgs &a = grp.p[0];
    if (n == 1 && act_mode == AQ3D_ACT_F32 && w_mode == AQ3D_W_F32 && cin <= 8 && kvol == 27 && cout == 16 && cap_in < (1 << 26)) {
        const dim3 grid(aq3d_div_up(cap_out, 256));
        switch (cin) {
#define AQ3D_STEMP(C_) case C_: hipLaunchKernelGGL(spconv_stem_pairs_kernel<C_>, grid, dim3(256), 0, stream, a); break;
            AQ3D_STEMP(1) AQ3D_STEMP(2) AQ3D_STEMP(3) AQ3D_STEMP(4) AQ3D_STEMP(5) AQ3D_STEMP(6) AQ3D_STEMP(7) AQ3D_STEMP(8)
#undef AQ3D_STEMP
        }
        AQ3D_CHECK_LAUNCH("aq3d_spconv(stem pairs)");
        return AQ3D_OK;
    }
    if (n == 1 && act_mode == AQ3D_ACT_F32 && w_mode == AQ3D_W_F32 && cin <= 8 && (cout & 3) == 0 && cout <= 64) {
        const dim3 grid(aq3d_div_up(cap_out, 256), cout / 4);
        switch (cin) {
#define AQ3D_STEM(C_) case C_: hipLaunchKernelGGL(spconv_stem_kernel<C_>, grid, dim3(256), 0, stream, a, cout); break;
            AQ3D_STEM(1) AQ3D_STEM(2) AQ3D_STEM(3) AQ3D_STEM(4) AQ3D_STEM(5) AQ3D_STEM(6) AQ3D_STEM(7) AQ3D_STEM(8)
#undef AQ3D_STEM
        }
        AQ3D_CHECK_LAUNCH("aq3d_spconv(stem)");
        return AQ3D_OK;
    }
    bool small_ok = act_mode == AQ3D_ACT_F32 && w_mode == AQ3D_W_F32 && cin == 128 && (kvol == 9 || kvol == 1) && !next_codes;
    for (int i = 0; i < n; ++i) small_ok = small_ok && grp.p[i].cout_real >= 1 && grp.p[i].cout_real <= 4 && !grp.p[i].residual;
    if (small_ok) {         // the head's output convs: a row per half wave on the plain vector ALU (see spconv_small_kernel)
        const dim3 grid(aq3d_grid(aq3d_div_up(cap_out, kvol == 1 ? 32 : 8), 1, 2048), n);
        if (kvol == 9) hipLaunchKernelGGL(spconv_small_kernel<9>, grid, dim3(256), 0, stream, grp, cout);
        else hipLaunchKernelGGL(spconv_small_kernel<1>, grid, dim3(256), 0, stream, grp, cout);
        AQ3D_CHECK_LAUNCH("aq3d_spconv(small)");
        return AQ3D_OK;
    }
    const int cin_pad = (cin + 15) / 16 * 16;
#define AQ3D_SPCONV_CASE(CP, CO, WR_, WC_) \
    if (cin_pad == CP && cout == CO) return spconv_launch_modes<CP, CO, WR_, WC_>(grp, n, act_mode, w_mode, stream);
    AQ3D_SPCONV_CASE(16, 16, 4, 1)
    AQ3D_SPCONV_CASE(16, 32, 2, 2)
    AQ3D_SPCONV_CASE(32, 32, 2, 2)
    AQ3D_SPCONV_CASE(32, 64, 2, 2)
    AQ3D_SPCONV_CASE(64, 64, 2, 2)
    AQ3D_SPCONV_CASE(64, 128, 2, 2)
    AQ3D_SPCONV_CASE(128, 128, 2, 2)
    AQ3D_SPCONV_CASE(128, 16, 4, 1)     // VoxelNeXt head output convs (1..3 channels padded to 16)
#undef AQ3D_SPCONV_CASE
    aq3d_set_error("aq3d_spconv: unsupported channel pair cin=%d (pad %d) cout=%d", cin, cin_pad, cout);
    return AQ3D_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------- per-column abs-max + quantizer
__global__ __launch_bounds__(256) void absmax_cols_kernel(const float *__restrict__ x, const int *count, int cap, int C,
                                                          float *amax) {
    __shared__ float red[256];
    int n = *count;
    if (n > cap) n = cap;
    // thread t owns column (t % C) of rows t / C, stepping by 256 / C rows (C divides 256 or C <= 256)
    const int rows_per_pass = 256 / C;
    const int c = threadIdx.x % C, r0 = threadIdx.x / C;
    float mx = 0.f;
    if (r0 < rows_per_pass)
        for (int r = blockIdx.x * rows_per_pass + r0; r < n; r += gridDim.x * rows_per_pass)
            mx = fmaxf(mx, fabsf(x[(size_t)r * C + c]));
    red[threadIdx.x] = mx;
    __syncthreads();
    if (threadIdx.x < C) {
        for (int j = 1; j < rows_per_pass; ++j) mx = fmaxf(mx, red[j * C + threadIdx.x]);
        if (mx > amax[threadIdx.x]) aq3d_atomic_max_pos(&amax[threadIdx.x], mx);
    }
}

extern "C" int aq3d_absmax_cols(const float *x, const int32_t *count, int32_t cap, int32_t C, float *amax,
                                void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(x && count && amax, "aq3d_absmax_cols: null argument");
    AQ3D_REQUIRE(C >= 1 && C <= 256, "aq3d_absmax_cols: C=%d not in 1..256", C);
    const int rows_per_pass = 256 / C;
    hipLaunchKernelGGL(absmax_cols_kernel, dim3(aq3d_grid(aq3d_div_up(cap, rows_per_pass), 1, 1024)), dim3(256), 0, stream,
                       x, count, cap, C, amax);
    AQ3D_CHECK_LAUNCH("aq3d_absmax_cols");
    return AQ3D_OK;
}

// TensorQuantizer restatement: scale = qmax / amax (0 where amax <= 2^-24), q = clamp(rint(x*scale), -qmax-1, qmax),
// fake-quant value = q / scale (scale := 1 where amax <= 2^-24).  rintf = round-half-to-even = torch.round.
__global__ __launch_bounds__(256) void quantize_rows_kernel(const float *__restrict__ x, const float *__restrict__ amax,
                                                            const int *count, int cap, int C, float qmax,
                                                            int per_tensor, int8_t *__restrict__ codes,
                                                            float *__restrict__ fq, float *__restrict__ dequant) {
    __shared__ float s_scale[256];
    __shared__ float s_div[256];
    int n = *count;
    if (n > cap) n = cap;
    if (threadIdx.x < C) {
        float am = amax[threadIdx.x];
        if (per_tensor) {
            am = 0.f;
            for (int c = 0; c < C; ++c) am = fmaxf(am, amax[c]);
        }
        const float eps = 1.0f / (float)(1 << 24);
        float scale = __fdiv_rn(qmax, am);
        float dv = scale;
        if (am <= eps) { scale = 0.f; dv = 1.f; }
        s_scale[threadIdx.x] = scale;
        s_div[threadIdx.x] = dv;
        if (dequant && blockIdx.x == 0) dequant[threadIdx.x] = __fdiv_rn(1.0f, dv);
    }
    __syncthreads();
    if ((C & 3) == 0 && (long long)cap * C < (1ll << 31)) {
        // four channels per thread: 128-bit loads, 64-bit f16 stores, 32-bit index arithmetic (the scalar loop below pays a
        // 64-bit division and modulo per element; this kernel runs ~20 times per frame between two sparse convs)
        const int C4 = C >> 2, total4 = n * C4;
        for (int t = blockIdx.x * 256 + threadIdx.x; t < total4; t += gridDim.x * 256) {
            const int r = t / C4, c = (t - r * C4) * 4;
            const float4 xv = reinterpret_cast<const float4 *>(x)[t];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            float v[4];
            signed char qc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float q = rintf(__fmul_rn(xs[k], s_scale[c + k]));
                q = fminf(fmaxf(q, -qmax - 1.f), qmax);
                qc[k] = (signed char)(int)q;
                v[k] = __fdiv_rn(q, s_div[c + k]);
            }
            if (codes) reinterpret_cast<char4 *>(codes)[t] = make_char4(qc[0], qc[1], qc[2], qc[3]);
            if (fq) reinterpret_cast<float4 *>(fq)[t] = make_float4(v[0], v[1], v[2], v[3]);
        }
        return;
    }
    const long long total = (long long)n * C;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        int c = (int)(t % C);
        float q = rintf(__fmul_rn(x[t], s_scale[c]));
        q = fminf(fmaxf(q, -qmax - 1.f), qmax);
        if (codes) codes[t] = (int8_t)(int)q;
        const float v = __fdiv_rn(q, s_div[c]);
        if (fq) fq[t] = v;
    }
}

extern "C" int aq3d_quantize_rows(const float *x, const float *amax, const int32_t *count, int32_t cap, int32_t C,
                                  int32_t bits, int32_t per_tensor, int8_t *codes_out, float *fq_out,
                                  float *dequant_out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(x && amax && count, "aq3d_quantize_rows: null argument");
    AQ3D_REQUIRE(C >= 1 && C <= 256, "aq3d_quantize_rows: C=%d not in 1..256", C);
    AQ3D_REQUIRE(bits >= 2 && bits <= 16, "aq3d_quantize_rows: bits=%d", bits);
    AQ3D_REQUIRE(!(codes_out && bits > 8), "aq3d_quantize_rows: int8 codes need bits <= 8");
    const float qmax = (float)((1 << (bits - 1)) - 1);
    hipLaunchKernelGGL(quantize_rows_kernel, dim3(aq3d_grid((int64_t)cap * C / ((C & 3) == 0 ? 4 : 1), 256)), dim3(256), 0, stream, x, amax, count,
                       cap, C, qmax, per_tensor, codes_out, fq_out, dequant_out);
    AQ3D_CHECK_LAUNCH("aq3d_quantize_rows");
    return AQ3D_OK;
}
