// Dense BEV / head convolutions as implicit GEMMs on the gfx950 matrix cores (NHWC, one frame per launch).
//
//   aq3d_conv2d_i8  — SQConv2d.forward (quant/smoothquant.py:38-99) without the im2col matrix:
//       the activation tile is divided by the SmoothQuant scale s[c,kh,kw], quantised per tensor and packed to
//       int8 WHILE it is staged into LDS (each tap has its own s, so each tap is quantised separately, exactly as
//       the reference's column-wise X / scale); weights arrive pre-smoothed and pre-quantised per frame from
//       aq3d_sq_prep; v_mfma_i32_16x16x64_i8 accumulates exactly in int32; the epilogue applies
//       (1/sx)(1/sw[oc]), bias, folded BatchNorm and ReLU.
//   aq3d_conv2d_f32 — the layers the reference leaves in fp32 (ConvTranspose2d deblocks, heat-map branch, the last
//       conv of every head branch: base_bev_backbone.py:48-66, center_head.py:22-29 + quant_centerpoint.py:28-71)
//       on the f32-input MFMA (bit-for-bit an fp32 fma chain).
//
// GEMM orientation: rows (MFMA A operand) = output channels, columns (B operand) = output pixels, so that each lane
// ends up with 4 consecutive output channels of one pixel → one 16-byte NHWC store per accumulator tile.
// Workgroup = 256 threads = 2x2 waves; wave tile = (TI*16) channels x (TJ*16) pixels.
#include "common.h"
#include "sq_layout.h"
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float *x;
    const void *w;          // i8 path: int8 [Cout][K];  f32 path: f32 [Cout_total][K]
    const float *s;         // i8: smoothing scale per column [K]
    const float *rinv;      // i8: 1/s
    const float *scal;      // i8: scal[0] = activation quant scale
    const float *out_mul;   // i8: per-oc dequant multiplier
    const float *bias, *post_scale, *post_shift;
    float *y;
    int H, W, Cin, Cout, ks, stride, pad, Ho, Wo, in_ld, out_ld, out_coff, deconv, relu;
    int rows_total;         // GEMM rows: Cout (conv) or deconv*deconv*Cout
    float *pieces_out;      // optional [16][pieces_ld] abs-max accumulator of the output image
    int pieces_ld;
};

__device__ __forceinline__ int piece_cls(int i, int n) { return i == 0 ? 0 : (i == n - 1 ? 1 : 2 + (i & 1)); }

// exact fp32 division x / s given r = RN(1/s): one Newton correction on the quotient (Markstein)
__device__ __forceinline__ float div_by_scale(float x, float s, float r) {
    float q0 = __fmul_rn(x, r);
    float e = __fmaf_rn(-q0, s, x);
    return __fmaf_rn(e, r, q0);
}

// Quantise 4 activations to int8 codes and pack them into one dword.
//   t = x / s (exact), u = t * sxv, code = rint(u) — rint through the 1.5*2^23 magic add (RNE by the adder, exact for
//   |u| < 2^22): the low byte of the sum's bit pattern is the two's-complement code. sxv is the activation scale, or 0 for
//   a pixel that falls into the zero padding (then every code is 0 whatever was loaded).
__device__ __forceinline__ int quant_pack4(f32x4 v, f32x4 s, f32x4 r, float sxv) {
    float u0 = __fmul_rn(div_by_scale(v.x, s.x, r.x), sxv);
    float u1 = __fmul_rn(div_by_scale(v.y, s.y, r.y), sxv);
    float u2 = __fmul_rn(div_by_scale(v.z, s.z, r.z), sxv);
    float u3 = __fmul_rn(div_by_scale(v.w, s.w, r.w), sxv);
    // clamp (one v_med3 each): only a calibrated (static) amax can actually be exceeded
    u0 = __builtin_amdgcn_fmed3f(u0, -128.f, 127.f); u1 = __builtin_amdgcn_fmed3f(u1, -128.f, 127.f);
    u2 = __builtin_amdgcn_fmed3f(u2, -128.f, 127.f); u3 = __builtin_amdgcn_fmed3f(u3, -128.f, 127.f);
    const float magic = 12582912.0f;
    const unsigned b0 = __float_as_uint(__fadd_rn(u0, magic)), b1 = __float_as_uint(__fadd_rn(u1, magic));
    const unsigned b2 = __float_as_uint(__fadd_rn(u2, magic)), b3 = __float_as_uint(__fadd_rn(u3, magic));
    const unsigned lo = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u);   // bytes: b0.0, b1.0, 0, 0
    const unsigned hi = __builtin_amdgcn_perm(b3, b2, 0x04000c0cu);   // bytes: 0, 0, b2.0, b3.0
    return (int)(lo | hi);
}

// ------------------------------------------------------------------ int8 SmoothQuant conv
template <int TI, int TJ>
__global__ __launch_bounds__(256) void conv2d_i8_kernel(ConvArgs a) {
    constexpr int BM = 32 * TI;   // output channels per workgroup
    constexpr int BN = 32 * TJ;   // pixels per workgroup
    constexpr int LD = 80;        // bytes per LDS row: 64 + 16 pad (conflict-free ds_read_b128 of 16 rows)
    constexpr int PPT = BN / 16;  // pixels per thread when staging X
    constexpr int WPT = BM / 64;  // 16-byte weight chunks per thread (BM*4 chunks / 256 threads)
    static_assert(BM >= 64, "BM >= 64");
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * LD];
    __shared__ unsigned int pm[16 * BM];     // piece maxima of this workgroup's outputs (bit patterns of |v|, v >= 0)
    auto Wbuf = [&](int b) -> char * { return smem + b * (BM + BN) * LD; };
    auto Xbuf = [&](int b) -> char * { return smem + b * (BM + BN) * LD + BM * LD; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int g = lane >> 4, l15 = lane & 15;
    const int L = a.Ho * a.Wo;
    const int K = a.Cin * a.ks * a.ks;
    const int pix0 = blockIdx.x * BN;
    const int oc0 = blockIdx.y * BM;
    const int cchunks = a.Cin >> 6;
    if (a.pieces_out)
        for (int i = tid; i < 16 * BM; i += 256) pm[i] = 0u;      // ordered before its use by the K loop's barriers
    const int nk = a.ks * a.ks * cchunks;
    const float sx = a.scal[0];

    // staging roles: thread = (channel quad cq, pixel group pg); per pixel the element offset of tap (0,0) and a 9-bit
    // mask of the taps that fall inside the image are computed ONCE; a K step only adds a wave-uniform tap offset.
    const int cq = tid & 15, pg = tid >> 4;
    int poff[PPT];
    unsigned pmask[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = pix0 + pg + 16 * i;
        poff[i] = 0;
        pmask[i] = 0u;
        if (p < L) {
            const int ho = p / a.Wo;
            const int h0 = ho * a.stride - a.pad, w0 = (p - ho * a.Wo) * a.stride - a.pad;
            poff[i] = (h0 * a.W + w0) * a.in_ld + cq * 4;
            for (int kh = 0; kh < a.ks; ++kh)
                for (int kw = 0; kw < a.ks; ++kw)
                    if (h0 + kh >= 0 && h0 + kh < a.H && w0 + kw >= 0 && w0 + kw < a.W) pmask[i] |= 1u << (kh * a.ks + kw);
        }
    }

    f32x4 xr[PPT];
    i32x4 wreg[WPT];
    f32x4 sr, rr;
    unsigned vbits = 0;   // validity of this step's PPT pixels (bit i)

    auto load_step = [&](int kstep) {
        const int tap = kstep / cchunks, c0 = (kstep - tap * cchunks) << 6;
        const int kh = tap / a.ks, kw = tap - kh * a.ks;
        const int toff = (kh * a.W + kw) * a.in_ld + c0;      // wave-uniform
        const int col = tap * a.Cin + c0 + cq * 4;
        sr = *reinterpret_cast<const f32x4 *>(a.s + col);
        rr = *reinterpret_cast<const f32x4 *>(a.rinv + col);
        vbits = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const bool ok = (pmask[i] >> tap) & 1u;
            vbits |= ok ? (1u << i) : 0u;
            const int off = ok ? poff[i] + toff : cq * 4;      // padding pixels read pixel 0 and are zeroed through sxv
            xr[i] = *reinterpret_cast<const f32x4 *>(a.x + off);
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int item = tid + 256 * i;        // (row, 16-byte chunk)
            const int row = min(oc0 + (item >> 2), a.Cout - 1), ch = item & 3;
            wreg[i] = *reinterpret_cast<const i32x4 *>(static_cast<const int8_t *>(a.w) + (size_t)row * K + (size_t)kstep * 64 + ch * 16);
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            *reinterpret_cast<int *>(Xbuf(buf) + (pg + 16 * i) * LD + cq * 4) =
                quant_pack4(xr[i], sr, rr, ((vbits >> i) & 1u) ? sx : 0.f);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int item = tid + 256 * i;
            *reinterpret_cast<i32x4 *>(Wbuf(buf) + (item >> 2) * LD + (item & 3) * 16) = wreg[i];
        }
    };

    i32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (i32x4){0, 0, 0, 0};

    auto mma_step = [&](int buf) {
        i32x4 af[TI], bf[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i)
            af[i] = *reinterpret_cast<const i32x4 *>(Wbuf(buf) + (wm * (BM / 2) + i * 16 + l15) * LD + g * 16);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
            bf[j] = *reinterpret_cast<const i32x4 *>(Xbuf(buf) + (wn * (BN / 2) + j * 16 + l15) * LD + g * 16);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    load_step(0);
    store_step(0);
    __syncthreads();
    // branch-free steady state (the last step is peeled): conditionals around the loads / stores inside the loop made the
    // compiler shuttle all accumulators between VGPRs and AGPRs every K step
    for (int k = 0; k + 1 < nk; ++k) {
        const int buf = k & 1;
        load_step(k + 1);
        mma_step(buf);
        store_step(buf ^ 1);
        __syncthreads();
    }
    mma_step((nk - 1) & 1);

    // epilogue: lane holds oc = base + 4g .. 4g+3 for pixel column l15
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int oc = oc0 + wm * (BM / 2) + i * 16 + 4 * g;
        if (oc >= a.Cout) continue;
        const f32x4 mul = *reinterpret_cast<const f32x4 *>(a.out_mul + oc);
        f32x4 b = (f32x4){0.f, 0.f, 0.f, 0.f}, ps = (f32x4){1.f, 1.f, 1.f, 1.f}, pt = b;
        if (a.bias) b = *reinterpret_cast<const f32x4 *>(a.bias + oc);
        if (a.post_scale) {
            ps = *reinterpret_cast<const f32x4 *>(a.post_scale + oc);
            pt = *reinterpret_cast<const f32x4 *>(a.post_shift + oc);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int p = pix0 + wn * (BN / 2) + j * 16 + l15;
            if (p >= L) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = __fmul_rn((float)acc[i][j][r], mul[r]);
                if (a.bias) t = __fadd_rn(t, b[r]);
                if (a.post_scale) t = __fadd_rn(__fmul_rn(t, ps[r]), pt[r]);
                if (a.relu) t = fmaxf(t, 0.f);
                v[r] = t;
            }
            *reinterpret_cast<f32x4 *>(a.y + (size_t)p * a.out_ld + a.out_coff + oc) = v;
            if (a.pieces_out) {
                const int ho = p / a.Wo;
                const int pc = piece_cls(ho, a.Ho) * 4 + piece_cls(p - ho * a.Wo, a.Wo);
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicMax(&pm[pc * BM + (oc - oc0) + r], __float_as_uint(fabsf(v[r])));
            }
        }
    }
    if (a.pieces_out) {
        __syncthreads();
        for (int i = tid; i < 16 * BM; i += 256) {
            const unsigned int v = pm[i];
            const int pc = i / BM, oc = oc0 + i % BM;
            if (v != 0u && oc < a.Cout) {
                unsigned int *dst = reinterpret_cast<unsigned int *>(a.pieces_out) + (size_t)pc * a.pieces_ld + a.out_coff + oc;
                if (v > *dst) atomicMax(dst, v);
            }
        }
    }
}

// ------------------------------------------------------------------ fp32 conv / transposed conv
// WM x WN waves; wave tile (TI*16) rows x (TJ*16) pixels; K step = 16 input channels of one tap.
template <int WM, int WN, int TI, int TJ>
__global__ __launch_bounds__(256) void conv2d_f32_kernel(ConvArgs a) {
    constexpr int BM = WM * TI * 16;
    constexpr int BN = WN * TJ * 16;
    constexpr int LD = 20;  // floats per LDS row: 16 + 4 pad
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LD];
    auto Wbuf = [&](int b) -> float * { return smem + b * (BM + BN) * LD; };
    auto Xbuf = [&](int b) -> float * { return smem + b * (BM + BN) * LD + BM * LD; };
    constexpr int XI = (BN * 4 + 255) / 256;  // float4 items per thread for X (BN rows x 4 quads)
    constexpr int WI = (BM * 4 + 255) / 256;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int g = lane >> 4, l15 = lane & 15;
    const int L = a.Ho * a.Wo;  // GEMM columns (for deconv Ho,Wo are the INPUT grid: one column per input pixel)
    const int K = a.Cin * a.ks * a.ks;
    const int pix0 = blockIdx.x * BN;
    const int row0 = blockIdx.y * BM;
    const int cchunks = a.Cin >> 4;
    const int nk = a.ks * a.ks * cchunks;

    int ph[XI], pw[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int item = tid + 256 * i;
        const int p = pix0 + (item >> 2);
        if ((item >> 2) < BN && p < L) {
            int ho = p / a.Wo;
            ph[i] = ho * a.stride - a.pad;
            pw[i] = (p - ho * a.Wo) * a.stride - a.pad;
        } else {
            ph[i] = -0x10000;
            pw[i] = -0x10000;
        }
    }
    f32x4 xr[XI], wr[WI];
    auto load_step = [&](int kstep) {
        const int tap = kstep / cchunks, c0 = (kstep - tap * cchunks) << 4;
        const int kh = tap / a.ks, kw = tap - kh * a.ks;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int item = tid + 256 * i;
            const int hi = ph[i] + kh, wi = pw[i] + kw;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W)
                v = *reinterpret_cast<const f32x4 *>(a.x + ((size_t)hi * a.W + wi) * a.in_ld + c0 + (item & 3) * 4);
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int item = tid + 256 * i;
            const int row = item >> 2;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (row < BM && row0 + row < a.rows_total)
                v = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(a.w) + (size_t)(row0 + row) * K +
                                                      (size_t)tap * a.Cin + c0 + (item & 3) * 4);
            wr[i] = v;
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int item = tid + 256 * i;
            if ((item >> 2) < BN) *reinterpret_cast<f32x4 *>(Xbuf(buf) + (item >> 2) * LD + (item & 3) * 4) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int item = tid + 256 * i;
            if ((item >> 2) < BM) *reinterpret_cast<f32x4 *>(Wbuf(buf) + (item >> 2) * LD + (item & 3) * 4) = wr[i];
        }
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto mma_step = [&](int buf) {
        f32x4 af[TI], bf[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i)
            af[i] = *reinterpret_cast<const f32x4 *>(Wbuf(buf) + (wm * TI * 16 + i * 16 + l15) * LD + g * 4);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
            bf[j] = *reinterpret_cast<const f32x4 *>(Xbuf(buf) + (wn * TJ * 16 + j * 16 + l15) * LD + g * 4);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
            }
    };
    load_step(0);
    store_step(0);
    __syncthreads();
    for (int k = 0; k + 1 < nk; ++k) {       // branch-free steady state, last step peeled (see conv2d_i8_kernel)
        const int buf = k & 1;
        load_step(k + 1);
        mma_step(buf);
        store_step(buf ^ 1);
        __syncthreads();
    }
    mma_step((nk - 1) & 1);

    const int d = a.deconv;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int rbase = row0 + wm * TI * 16 + i * 16 + 4 * g;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int p = pix0 + wn * TJ * 16 + j * 16 + l15;
            if (p >= L) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rbase + r;
                if (row >= a.rows_total) continue;
                int oc = row;
                size_t opix = (size_t)p;
                if (d > 0) {  // row = (i*d + j)*Cout + oc ; input pixel p = (h, w) → output (h*d+i, w*d+j)
                    const int ij = row / a.Cout;
                    oc = row - ij * a.Cout;
                    const int h = p / a.Wo, w = p - h * a.Wo;
                    opix = (size_t)(h * d + ij / d) * (a.Wo * d) + (w * d + ij % d);
                }
                float t = acc[i][j][r];
                if (a.bias) t = __fadd_rn(t, a.bias[oc]);
                if (a.post_scale) t = __fadd_rn(__fmul_rn(t, a.post_scale[oc]), a.post_shift[oc]);
                if (a.relu) t = fmaxf(t, 0.f);
                a.y[opix * a.out_ld + a.out_coff + oc] = t;
                if (a.pieces_out) {
                    const int OW = d > 0 ? a.Wo * d : a.Wo, OH = d > 0 ? a.Ho * d : a.Ho;
                    const int oh = (int)(opix / OW), ow = (int)(opix - (size_t)oh * OW);
                    unsigned int *dst = reinterpret_cast<unsigned int *>(a.pieces_out) +
                                        (size_t)(piece_cls(oh, OH) * 4 + piece_cls(ow, OW)) * a.pieces_ld + a.out_coff + oc;
                    const unsigned int vb = __float_as_uint(fabsf(t));
                    if (vb > *dst) atomicMax(dst, vb);
                }
            }
        }
    }
}

static int fill_args(ConvArgs *a, const float *x, const aq3d_conv2d_geom *g, const float *bias, const float *ps,
                     const float *pt, float *y, const char *who) {
    AQ3D_REQUIRE(x && g && y, "%s: null argument", who);
    AQ3D_REQUIRE((ps == nullptr) == (pt == nullptr), "%s: post_scale/post_shift must come together", who);
    AQ3D_REQUIRE(g->ksize == 1 || g->ksize == 3, "%s: ksize=%d", who, g->ksize);
    AQ3D_REQUIRE(g->stride >= 1 && g->pad >= 0 && g->H > 0 && g->W > 0 && g->Ho > 0 && g->Wo > 0, "%s: bad geometry", who);
    AQ3D_REQUIRE((g->in_ld & 3) == 0 && g->in_ld >= g->Cin, "%s: in_ld=%d", who, g->in_ld);
    a->x = x; a->bias = bias; a->post_scale = ps; a->post_shift = pt; a->y = y;
    a->H = g->H; a->W = g->W; a->Cin = g->Cin; a->Cout = g->Cout; a->ks = g->ksize; a->stride = g->stride; a->pad = g->pad;
    a->Ho = g->Ho; a->Wo = g->Wo; a->in_ld = g->in_ld; a->out_ld = g->out_ld; a->out_coff = g->out_coff;
    a->deconv = g->deconv; a->relu = g->relu;
    a->s = a->rinv = a->scal = a->out_mul = nullptr;
    a->pieces_out = nullptr;
    a->pieces_ld = 0;
    a->w = nullptr;
    a->rows_total = g->Cout;
    return AQ3D_OK;
}

extern "C" int aq3d_conv2d_i8(const float *x, const void *params, const aq3d_conv2d_geom *g, const float *bias,
                              const float *post_scale, const float *post_shift, float *y, float *pieces_out,
                              int32_t pieces_ld, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ConvArgs a;
    int rc = fill_args(&a, x, g, bias, post_scale, post_shift, y, "aq3d_conv2d_i8");
    if (rc != AQ3D_OK) return rc;
    AQ3D_REQUIRE(!pieces_out || (pieces_ld >= g->out_coff + g->Cout && g->Ho >= 3 && g->Wo >= 3), "aq3d_conv2d_i8: bad pieces_out layout");
    a.pieces_out = pieces_out;
    a.pieces_ld = pieces_ld;
    AQ3D_REQUIRE(params, "aq3d_conv2d_i8: null params");
    AQ3D_REQUIRE((g->Cin & 63) == 0, "aq3d_conv2d_i8: Cin=%d must be a multiple of 64", g->Cin);
    AQ3D_REQUIRE((g->Cout & 3) == 0 && (g->out_ld & 3) == 0 && (g->out_coff & 3) == 0, "aq3d_conv2d_i8: Cout/out_ld/out_coff must be multiples of 4");
    AQ3D_REQUIRE(g->deconv == 0, "aq3d_conv2d_i8: deconv unsupported");
    AQ3D_REQUIRE(g->Ho == (g->H + 2 * g->pad - g->ksize) / g->stride + 1 && g->Wo == (g->W + 2 * g->pad - g->ksize) / g->stride + 1,
                 "aq3d_conv2d_i8: Ho/Wo inconsistent with geometry");
    const SqLayout l = sq_layout(g->Cin, g->Cout, g->ksize);
    const char *base = (const char *)params;
    a.s = (const float *)(base + l.s);
    a.rinv = (const float *)(base + l.r);
    a.out_mul = (const float *)(base + l.mul);
    a.scal = (const float *)(base + l.scal);
    a.w = base + l.wq;
    const int L = g->Ho * g->Wo;
    // pixel-tile width: enough workgroups for >= 2 per CU (the K loop is a latency chain, it needs co-resident partners)
    static const int force_bn = getenv("AQ3D_I8_BN") ? atoi(getenv("AQ3D_I8_BN")) : 0;
    const int ocb = g->Cout >= 128 ? aq3d_div_up(g->Cout, 128) : aq3d_div_up(g->Cout, 64);
    int bn = 128;
    if (aq3d_div_up(L, 128) * ocb < 512) bn = 64;
    if (aq3d_div_up(L, 64) * ocb < 512) bn = 32;
    if (force_bn) bn = force_bn;
#define AQ3D_I8_LAUNCH(TI_, TJ_) \
    hipLaunchKernelGGL((conv2d_i8_kernel<TI_, TJ_>), dim3(aq3d_div_up(L, 32 * TJ_), ocb), dim3(256), 0, stream, a)
    if (g->Cout >= 128) {
        if (bn == 128) AQ3D_I8_LAUNCH(4, 4);
        else if (bn == 64) AQ3D_I8_LAUNCH(4, 2);
        else AQ3D_I8_LAUNCH(4, 1);
    } else {
        if (bn == 128) AQ3D_I8_LAUNCH(2, 4);
        else if (bn == 64) AQ3D_I8_LAUNCH(2, 2);
        else AQ3D_I8_LAUNCH(2, 1);
    }
#undef AQ3D_I8_LAUNCH
    AQ3D_CHECK_LAUNCH("aq3d_conv2d_i8");
    return AQ3D_OK;
}

extern "C" int aq3d_conv2d_f32(const float *x, const float *weight, const aq3d_conv2d_geom *g, const float *bias,
                               const float *post_scale, const float *post_shift, float *y, float *pieces_out,
                               int32_t pieces_ld, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ConvArgs a;
    int rc = fill_args(&a, x, g, bias, post_scale, post_shift, y, "aq3d_conv2d_f32");
    if (rc != AQ3D_OK) return rc;
    AQ3D_REQUIRE(!pieces_out || pieces_ld >= g->out_coff + g->Cout, "aq3d_conv2d_f32: bad pieces_out layout");
    a.pieces_out = pieces_out;
    a.pieces_ld = pieces_ld;
    AQ3D_REQUIRE(weight, "aq3d_conv2d_f32: null weight");
    AQ3D_REQUIRE((g->Cin & 15) == 0, "aq3d_conv2d_f32: Cin=%d must be a multiple of 16", g->Cin);
    a.w = weight;
    int L;
    if (g->deconv > 0) {
        AQ3D_REQUIRE(g->ksize == 1 && g->stride == 1 && g->pad == 0, "aq3d_conv2d_f32: deconv needs ksize=1 stride=1 pad=0 (kernel == stride == deconv)");
        AQ3D_REQUIRE(g->Ho == g->H && g->Wo == g->W, "aq3d_conv2d_f32: deconv takes Ho=H, Wo=W (the input grid)");
        a.rows_total = g->deconv * g->deconv * g->Cout;
        L = g->H * g->W;
    } else {
        AQ3D_REQUIRE(g->Ho == (g->H + 2 * g->pad - g->ksize) / g->stride + 1 && g->Wo == (g->W + 2 * g->pad - g->ksize) / g->stride + 1,
                     "aq3d_conv2d_f32: Ho/Wo inconsistent with geometry");
        L = g->Ho * g->Wo;
    }
    const int rows = a.rows_total;
    if (rows <= 16) {
        hipLaunchKernelGGL((conv2d_f32_kernel<1, 4, 1, 2>), dim3(aq3d_div_up(L, 128), 1), dim3(256), 0, stream, a);
    } else if (rows <= 64 || L < 128 * 150) {
        hipLaunchKernelGGL((conv2d_f32_kernel<2, 2, 2, 2>), dim3(aq3d_div_up(L, 64), aq3d_div_up(rows, 64)), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL((conv2d_f32_kernel<2, 2, 4, 2>), dim3(aq3d_div_up(L, 64), aq3d_div_up(rows, 128)), dim3(256), 0, stream, a);
    }
    AQ3D_CHECK_LAUNCH("aq3d_conv2d_f32");
    return AQ3D_OK;
}
