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
#include <stdio.h>
#include <type_traits>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float *x;
    const void *w;          // i8 path: int8 [Cout][K];  f32 path: f32 [Cout_total][K]
    const void *w16;        // f32 path, optional: the same weights split into fp16 planes [Cout_total][2][K] (aq3d_conv2d_split_weight)
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
    int pieces_mode;        // 0: 16 exact classes; 1: the four interior classes merged (all consumers have stride 1)
    int xcd_order;          // int8 generic kernel: pixel tiles dealt to XCDs in contiguous ranges (grid x padded to 8)
    int nrb;                // fp16-pipe fp32 kernel: row (output-channel) blocks per pixel tile (1-D XCD-aware tile order)
};

// Up to 8 independent convolutions of one launch class in a single grid (blockIdx.z = problem): the head's per-branch convs
// are each too small to fill the chip and were a third of the dense launches.
#define AQ3D_MAX_GROUP 8
struct ConvGroup {
    ConvArgs p[AQ3D_MAX_GROUP];
};

__device__ __forceinline__ int piece_cls(int i, int n) { return i == 0 ? 0 : (i == n - 1 ? 1 : 2 + (i & 1)); }

// exact fp32 division x / s given r = RN(1/s): one Newton correction on the quotient (Markstein)
__device__ __forceinline__ float div_by_scale(float x, float s, float r) {
    float q0 = __fmul_rn(x, r);
    float e = __fmaf_rn(-q0, s, x);
    return __fmaf_rn(e, r, q0);
}

// Quantise 4 activations to int8 codes and pack them into one dword.
//   t = x / s (exact: Markstein correction of x * RN(1/s)), u = t * sxv, code = rint(u) — rint through the 1.5*2^23 magic
//   add (RNE by the adder, exact for |u| < 2^22): the low byte of the sum's bit pattern is the two's-complement code. sxv is
//   the activation scale, or 0 for a pixel that falls into the zero padding (then every code is 0 whatever was loaded).
//   Scalar fp32 ops on purpose: the packed forms (v_pk_fma_f32 ...) measured ~4x the issue cost of a plain VALU op on
//   gfx950 — in-kernel stamps: 40 packed + 44 plain ops took ~1000 cycles per wave. CLAMP is only needed with a calibrated
//   (static) amax; a dynamic per-tensor amax bounds |u| by 127 (1 + 2^-23), which rounds to 127.
template <bool CLAMP>
__device__ __forceinline__ int quant_pack4(f32x4 v, f32x4 s, f32x4 r, float sxv) {
    float u0 = __fmul_rn(div_by_scale(v.x, s.x, r.x), sxv);
    float u1 = __fmul_rn(div_by_scale(v.y, s.y, r.y), sxv);
    float u2 = __fmul_rn(div_by_scale(v.z, s.z, r.z), sxv);
    float u3 = __fmul_rn(div_by_scale(v.w, s.w, r.w), sxv);
    if (CLAMP) {
        u0 = __builtin_amdgcn_fmed3f(u0, -128.f, 127.f); u1 = __builtin_amdgcn_fmed3f(u1, -128.f, 127.f);
        u2 = __builtin_amdgcn_fmed3f(u2, -128.f, 127.f); u3 = __builtin_amdgcn_fmed3f(u3, -128.f, 127.f);
    }
    const float magic = 12582912.0f;
    const unsigned b0 = __float_as_uint(__fadd_rn(u0, magic)), b1 = __float_as_uint(__fadd_rn(u1, magic));
    const unsigned b2 = __float_as_uint(__fadd_rn(u2, magic)), b3 = __float_as_uint(__fadd_rn(u3, magic));
    const unsigned lo = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u);   // bytes: b0.0, b1.0, 0, 0
    const unsigned hi = __builtin_amdgcn_perm(b3, b2, 0x04000c0cu);   // bytes: 0, 0, b2.0, b3.0
    return (int)(lo | hi);
}

// In-kernel cycle stamps of conv2d_i8_kernel for scratch/cw_stamps.py --dense (diagnostic build with -DAQ3D_STAMPS only).
#ifdef AQ3D_STAMPS
__device__ unsigned long long *g_i8_stamps;          // [workgroups][waves][8]
extern "C" int aq3d_diag_set_i8_stamps(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_i8_stamps), &p, sizeof(p)); }
#define I8_STAMP(var) const unsigned long long var = (unsigned long long)clock64()
#define I8_ACC(acc, from, to) acc += (to) - (from)
#else
#define I8_STAMP(var)
#define I8_ACC(acc, from, to)
#endif
// Knock-out build of conv2d_i8_kernel for scratch/i8_knockout.py (-DAQ3D_KO=mask; timing only, results are wrong; the product
// build compiles none of it). bit 0: s | 1/s not loaded; 1: X loaded for the first step only; 2: W loaded for the first step only;
// 3: no quantisation arithmetic; 4: one MFMA of eight; 5: no LDS stores of the codes. Every workgroup records where and when it ran.
#ifdef AQ3D_KO
#define AQ3D_KO_PLACE 1
__device__ unsigned long long *g_i8_place;           // [workgroups][4]: XCC id, HW id, start, end (100 MHz wall clock)
extern "C" int aq3d_diag_set_i8_place(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_i8_place), &p, sizeof(p)); }
#else
#define AQ3D_KO 0
#endif
#ifndef AQ3D_EXP
#define AQ3D_EXP 0       // A/B switches of diagnostic builds (scratch/ab_build.py): 4 = non-temporal output stores, 8 = no row-register kernel
#endif

// ------------------------------------------------------------------ int8 SmoothQuant conv: epilogue of both kernels
// A lane holds the 4 consecutive output channels base + 4g .. 4g+3 of pixel column l15 of each of its TI x TJ accumulator tiles;
// pixw = first pixel of the wave's columns, ep = {out_mul, bias, post_scale, post_shift} of the block's channels (LDS), pm = the
// workgroup's piece maxima (LDS, zeroed at kernel start when a.pieces_out is set).
template <int TI, int TJ, int BM, int NTH>
__device__ __forceinline__ void i8_epilogue(const ConvArgs &a, i32x4 (&acc)[TI][TJ], unsigned int *pm, float (*ep)[BM], int pixw, int oc0,
                                            int wm, int L) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = lane >> 4, l15 = lane & 15;
    // Order (stamps, scratch/cw_stamps.py --dense: the epilogue was
    // 28 % of the 128 -> 128 layer): operands from LDS (staged at kernel start), values into registers, the output's piece maxima
    // (interior: DPP row reduction over the 16 pixel lanes, no LDS shuffles), barrier + flush, and the 16-byte output stores LAST —
    // nothing waits for them.
    int pcls[TJ];                           // piece class of this lane's TJ pixels; -1 = interior in merged mode
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        pcls[j] = 0;
        if (a.pieces_out) {
            const int p = pixw + j * 16 + l15;
            const int ho = p / a.Wo, wo = p - ho * a.Wo;
            pcls[j] = (a.pieces_mode == 1 && ho > 0 && ho < a.Ho - 1 && wo > 0 && wo < a.Wo - 1)
                          ? -1 : piece_cls(ho, a.Ho) * 4 + piece_cls(wo, a.Wo);
        }
    }
    f32x4 vout[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int ol = wm * (BM / 2) + i * 16 + 4 * g;       // channel inside the workgroup's block
        const int oc = oc0 + ol;
        const f32x4 mul = *reinterpret_cast<const f32x4 *>(&ep[0][ol]), b = *reinterpret_cast<const f32x4 *>(&ep[1][ol]);
        const f32x4 ps = *reinterpret_cast<const f32x4 *>(&ep[2][ol]), pt = *reinterpret_cast<const f32x4 *>(&ep[3][ol]);
        float mint[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int p = pixw + j * 16 + l15;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = __fmul_rn((float)acc[i][j][r], mul[r]);
                if (a.bias) t = __fadd_rn(t, b[r]);
                if (a.post_scale) t = __fadd_rn(__fmul_rn(t, ps[r]), pt[r]);
                if (a.relu) t = fmaxf(t, 0.f);
                v[r] = t;
            }
            vout[i][j] = v;
            if (a.pieces_out && p < L && oc < a.Cout) {
                if (pcls[j] < 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) mint[r] = fmaxf(mint[r], fabsf(v[r]));      // interior: reduced in registers below
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicMax(&pm[pcls[j] * BM + ol + r], __float_as_uint(fabsf(v[r])));
                }
            }
        }
        if (a.pieces_out && a.pieces_mode == 1) {
            // interior pixels (98 % of the image): max over this wave's 16 pixel lanes = one DPP row (lanes 16g .. 16g+15), four
            // row_shr steps leave it in the row's last lane, then ONE LDS atomic per (wave, channel). Values are >= 0: the zeros
            // DPP feeds for lanes without a source are neutral.
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m = mint[r];
                m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x111, 0xf, 0xf, true)));
                m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x112, 0xf, 0xf, true)));
                m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x114, 0xf, 0xf, true)));
                m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x118, 0xf, 0xf, true)));
                if (l15 == 15 && m > 0.f) atomicMax(&pm[10 * BM + ol + r], __float_as_uint(m));
            }
        }
    }
    if (a.pieces_out) {
        __syncthreads();
        for (int i = tid; i < 16 * BM; i += NTH) {
            const unsigned int v = pm[i];
            const int pc = i / BM, oc = oc0 + i % BM;
            if (v != 0u && oc < a.Cout) {
                unsigned int *dst = reinterpret_cast<unsigned int *>(a.pieces_out) + (size_t)pc * a.pieces_ld + a.out_coff + oc;
                // mode 1: the merged interior maximum may land in ANY of the four interior classes — every stride-1 consumer takes
                // the max over all four — so workgroups spread over them: same-address global atomics serialise in L2 (~50 ns
                // each: 552 workgroups on one slot row lengthen the kernel's tail by tens of microseconds). Fire-and-forget,
                // no dependent read.
                if (a.pieces_mode == 1) {
                    const int alt[4] = {10, 11, 14, 15};
                    const int pc2 = pc == 10 ? alt[blockIdx.x & 3] : pc;
                    atomicMax(reinterpret_cast<unsigned int *>(a.pieces_out) + (size_t)pc2 * a.pieces_ld + a.out_coff + oc, v);
                } else if (v > *dst) atomicMax(dst, v);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int oc = oc0 + wm * (BM / 2) + i * 16 + 4 * g;
        if (oc >= a.Cout) continue;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int p = pixw + j * 16 + l15;
#if AQ3D_EXP & 4
            if (p < L) __builtin_nontemporal_store(vout[i][j], reinterpret_cast<f32x4 *>(a.y + (size_t)p * a.out_ld + a.out_coff + oc));
#else
            if (p < L) *reinterpret_cast<f32x4 *>(a.y + (size_t)p * a.out_ld + a.out_coff + oc) = vout[i][j];
#endif
        }
    }
}

// ------------------------------------------------------------------ int8 SmoothQuant conv
// NWN = wave columns (2: 256 threads, 4: 512 threads). The K loop is a per-step dependency chain (global load -> quantise ->
// LDS -> barrier -> MFMA); 8 waves split the staging work of a step in half and give every SIMD a second wave to interleave.
template <int TI, int TJ, int NWN>
__global__ __launch_bounds__(128 * NWN) void conv2d_i8_kernel(ConvGroup grp) {
    constexpr int NTH = 128 * NWN;
    const ConvArgs a = grp.p[blockIdx.z];
    if ((int)blockIdx.y * 32 * TI >= a.Cout) return;      // a problem with fewer channel blocks than the widest of its group
    constexpr int BM = 32 * TI;   // output channels per workgroup
    constexpr int BN = 16 * TJ * NWN;   // pixels per workgroup
    // 64-byte LDS rows (one K step of a channel / pixel), the four 16-byte chunks of row r stored at chunk position c ^ f(r),
    // f = {0, 2, 3, 1}[(r >> 2) & 3]: ds_read_b128 serves a wave in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... and
    // with this f the 16 rows x chunks of every group fall on 16 different 16-byte bank slots. (The 80-byte padded rows of rounds 1-2
    // were conflict-free for 16 consecutive lanes only: PMC counted 44 % conflict cycles.)
    constexpr int LD = 64;
    auto swz = [](int row, int chunk) { return chunk ^ ((0x1320 >> (((row >> 2) & 3) * 4)) & 3); };
    constexpr int PGS = NTH / 16; // pixel groups of the staging role (16 channel quads x PGS pixels per pass)
    constexpr int PPT = BN / PGS; // pixels per thread when staging X
    constexpr int WPT = (BM * 4 + NTH - 1) / NTH;  // 16-byte weight chunks per thread
    static_assert(BM >= 64, "BM >= 64");
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * LD];
    __shared__ unsigned int pm[16 * BM];     // piece maxima of this workgroup's outputs (bit patterns of |v|, v >= 0)
    __shared__ __attribute__((aligned(16))) float ep[4][BM];   // epilogue operands of the block's channels: out_mul, bias, post_scale, post_shift
    auto Wbuf = [&](int b) -> char * { return smem + b * (BM + BN) * LD; };
    auto Xbuf = [&](int b) -> char * { return smem + b * (BM + BN) * LD + BM * LD; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int g = lane >> 4, l15 = lane & 15;
    const int L = a.Ho * a.Wo;
    const int K = a.Cin * a.ks * a.ks;
    // XCD-aware tile order: workgroups go round-robin over the 8 XCDs (each with its own L2), so consecutive block ids would
    // scatter a 3x3 conv's vertically adjacent pixel tiles — which re-read each other's input rows — over all eight L2s.
    // gridDim.x is a multiple of 8 (host): XCD x takes the contiguous pixel range [x, x+1) * gridDim.x/8 tiles.
    const int bx = a.xcd_order ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int pix0 = bx * BN;
    if (pix0 >= L) return;
    I8_STAMP(t_start);
#ifdef AQ3D_STAMPS
    unsigned long long s_load = 0, s_mma = 0, s_store = 0, s_bar = 0;
#endif
#ifdef AQ3D_KO_PLACE
    const unsigned long long ko_t0 = wall_clock64();
#endif
    const int oc0 = blockIdx.y * BM;
    const int cchunks = a.Cin >> 6;
    if (a.pieces_out)
        for (int i = tid; i < 16 * BM; i += NTH) pm[i] = 0u;      // ordered before its use by the K loop's barriers
    for (int i = tid; i < BM; i += NTH) {                          // read back after the K loop (its barriers order the two)
        const int oc = oc0 + i;
        const bool in = oc < a.Cout;
        ep[0][i] = in ? a.out_mul[oc] : 0.f;
        ep[1][i] = (in && a.bias) ? a.bias[oc] : 0.f;
        ep[2][i] = (in && a.post_scale) ? a.post_scale[oc] : 1.f;
        ep[3][i] = (in && a.post_scale) ? a.post_shift[oc] : 0.f;
    }
    const int ntaps = a.ks * a.ks;
    const int nk = ntaps * cchunks;
    const float sx = a.scal[0];
    const bool clamp = a.scal[3] != 0.f;     // calibrated (static) activation amax: codes can exceed +-127 before the clamp

    // staging roles: thread = (channel quad cq, pixel group pg); per pixel the element offset of tap (0,0) and a 9-bit
    // mask of the taps that fall inside the image are computed ONCE; a K step only adds a wave-uniform tap offset.
    const int cq = tid & 15, pg = tid >> 4;
    int poff[PPT];
    unsigned pmask[PPT];
    int ho_i = (pix0 + pg) / a.Wo, wo_i = (pix0 + pg) - ho_i * a.Wo;      // one division; the thread's pixels are PGS apart
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = pix0 + pg + PGS * i;
        poff[i] = 0;
        pmask[i] = 0u;
        const int ho = ho_i, wo = wo_i;
        wo_i += PGS;
        while (wo_i >= a.Wo) { wo_i -= a.Wo; ++ho_i; }
        if (p < L) {
            const int h0 = ho * a.stride - a.pad, w0 = wo * a.stride - a.pad;
            poff[i] = (h0 * a.W + w0) * a.in_ld + cq * 4;
            unsigned hm = 0u, wm = 0u;              // rows / columns of the window inside the image; mask = their outer product
            for (int k = 0; k < a.ks; ++k) {
                hm |= (unsigned)((unsigned)(h0 + k) < (unsigned)a.H) << k;
                wm |= (unsigned)((unsigned)(w0 + k) < (unsigned)a.W) << k;
            }
            for (int kh = 0; kh < a.ks; ++kh)
                if ((hm >> kh) & 1u) pmask[i] |= wm << (kh * a.ks);
        }
    }

    struct Stage {                  // registers of the K step in flight
        f32x4 xr[PPT];
        i32x4 wreg[WPT];
        f32x4 sr, rr;
    };
    Stage stA;
    // every global read of the K loop is a bounded buffer load with a 32-bit byte offset: a tap that falls into the zero
    // padding is an out-of-range offset (returns zeros -> code 0), so there is no validity bookkeeping, no select of the
    // activation scale and no 64-bit address arithmetic per load (they were a third of the loop's VALU instructions)
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.H * a.W * a.in_ld * 4, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.Cout * K, 0x00020000);
    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void *)a.s, 0, K * 4, 0x00020000);
    const auto rsrc_r = __builtin_amdgcn_make_buffer_rsrc((void *)a.rinv, 0, K * 4, 0x00020000);
    int wrow_off[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int item = min(tid + NTH * i, BM * 4 - 1);        // (row, 16-byte chunk)
        wrow_off[i] = min(oc0 + (item >> 2), a.Cout - 1) * K + (item & 3) * 16;
    }

    // K steps are issued strictly in order 0, 1, 2, ...: (tap, kh, kw, channel chunk) advance as counters (two integer divisions by
    // run-time values per step were a visible part of the 380 cycles a wave spent ISSUING a step's loads, scratch/cw_stamps.py)
    int ls_tap = 0, ls_kh = 0, ls_kw = 0, ls_c0 = 0;
    auto load_step = [&](Stage &st, int step) {
        // K order: channel chunk outermost, the taps of one chunk back to back — the nine taps re-read the same fp32 pixels
        // (shifted by a pixel or a row), so their re-use distance is one step (16 KB per workgroup) instead of Cin/64 steps:
        // with tap-major order the 512-channel head conv fetched its input 8.6x from beyond L2. int32 sums: order-free.
        const int tap = ls_tap, c0 = ls_c0;
        const int toff = (ls_kh * a.W + ls_kw) * a.in_ld + c0;      // wave-uniform
        if (++ls_kw == a.ks) { ls_kw = 0; ++ls_kh; }
        if (++ls_tap == ntaps) { ls_tap = 0; ls_kh = 0; ls_kw = 0; ls_c0 += 64; }
        // (measured round 3: fetching s | 1/s with lanes 0-15 only and handing them to the other three pixel groups of the wave by
        // ds_bpermute — a quarter of the vector-memory bytes of these two loads — is 4-11 % SLOWER: the loads hit L1, the shuffles wait)
        const unsigned col = (unsigned)(tap * a.Cin + c0 + cq * 4) * 4u;
        if (!(AQ3D_KO & 1) || step == 0) {
            st.sr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, col, 0, 0));
            st.rr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, col, 0, 0));
        }
        if (!(AQ3D_KO & 2) || step == 0) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const bool ok = (pmask[i] >> tap) & 1u;
                const unsigned off = ok ? (unsigned)(poff[i] + toff) * 4u : 0xfffffff0u;
                st.xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, off, 0, 0));
            }
        }
        if (!(AQ3D_KO & 4) || step == 0) {
#pragma unroll
            for (int i = 0; i < WPT; ++i)
                st.wreg[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)(wrow_off[i] + tap * a.Cin + c0), 0, 0));
        }
    };
    auto store_step = [&](auto clamp_c, const Stage &st, int buf) {
        constexpr bool CL = decltype(clamp_c)::value;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            int code;
            if (AQ3D_KO & 8) code = __float_as_int(st.xr[i].x) + __float_as_int(st.xr[i].y) + __float_as_int(st.xr[i].z) + __float_as_int(st.xr[i].w) + __float_as_int(st.sr.x);
            else code = quant_pack4<CL>(st.xr[i], st.sr, st.rr, sx);
            if (AQ3D_KO & 32) asm volatile("" ::"v"(code));
            else *reinterpret_cast<int *>(Xbuf(buf) + (pg + PGS * i) * LD + swz(pg + PGS * i, cq >> 2) * 16 + (cq & 3) * 4) = code;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int item = tid + NTH * i;
            if (BM * 4 % NTH == 0 || item < BM * 4)
                *reinterpret_cast<i32x4 *>(Wbuf(buf) + (item >> 2) * LD + swz(item >> 2, item & 3) * 16) = st.wreg[i];
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
            af[i] = *reinterpret_cast<const i32x4 *>(Wbuf(buf) + (wm * (BM / 2) + i * 16 + l15) * LD + swz(l15, g) * 16);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
            bf[j] = *reinterpret_cast<const i32x4 *>(Xbuf(buf) + (wn * (BN / NWN) + j * 16 + l15) * LD + swz(l15, g) * 16);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                if ((AQ3D_KO & 16) && (i | j)) asm volatile("" ::"v"(af[i]), "v"(bf[j]));
                else acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
    };
    // The whole K loop exists twice, with and without the clamp of calibrated (static) scales: a wave-uniform branch INSIDE the loop
    // made the register allocator rotate the accumulators through other AGPRs every step (40 v_accvgpr_* per step on top of ~110
    // useful VALU instructions, each waiting for the MFMA that produced its source).
    I8_STAMP(t_pro);
#ifdef AQ3D_STAMPS
    unsigned long long t_fill_v = 0;
#endif
    load_step(stA, 0);          // issued before `clamp` (a global load itself) is needed to pick the loop variant
    auto k_loop = [&](auto clamp_c) {
        store_step(clamp_c, stA, 0);
        __syncthreads();
#ifdef AQ3D_STAMPS
        t_fill_v = (unsigned long long)clock64();
#endif
        // branch-free steady state (the last step is peeled): conditionals around the loads / stores inside the loop made the
        // compiler shuttle all accumulators between VGPRs and AGPRs every K step
        for (int k = 0; k + 1 < nk; ++k) {
            const int buf = k & 1;
            I8_STAMP(ta);
            load_step(stA, k + 1);
            __builtin_amdgcn_sched_barrier(0);
            I8_STAMP(tb);
            mma_step(buf);
            __builtin_amdgcn_sched_barrier(0);
            I8_STAMP(tc);
            store_step(clamp_c, stA, buf ^ 1);
            I8_STAMP(td);
            __syncthreads();
            I8_STAMP(te);
            I8_ACC(s_load, ta, tb); I8_ACC(s_mma, tb, tc); I8_ACC(s_store, tc, td); I8_ACC(s_bar, td, te);
        }
        mma_step((nk - 1) & 1);
    };
    if (clamp) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    I8_STAMP(t_loop);

    i8_epilogue<TI, TJ, BM, NTH>(a, acc, pm, ep, pix0 + wn * (BN / NWN), oc0, wm, L);
#ifdef AQ3D_STAMPS
    if (lane == 0 && g_i8_stamps) {
        unsigned long long *o = g_i8_stamps + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (NTH / 64) + wave) * 8;
        const unsigned long long t_end = (unsigned long long)clock64();
        o[0] = t_pro - t_start; o[1] = t_fill_v - t_pro; o[2] = s_load; o[3] = s_mma; o[4] = s_store; o[5] = s_bar;
        o[6] = t_end - t_loop; o[7] = ((unsigned long long)nk << 48) | (t_end - t_start);
    }
#endif
#ifdef AQ3D_KO_PLACE
    if (tid == 0 && g_i8_place) {
        unsigned long long *o = g_i8_place + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4;
        o[0] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        o[1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
        o[2] = ko_t0;
        o[3] = wall_clock64();
    }
#endif
}

// ------------------------------------------------------------------ int8 SmoothQuant conv, 3x3 / stride 1 / pad 1: row-register staging
// Same tiles, K order, LDS layout, MFMA loop and epilogue as conv2d_i8_kernel; what differs is how a K step's operands reach the
// quantiser. Knock-out builds of the kernel above (scratch/i8_knockout.py, profiles/r04/i8_knockout.jsonl) put the
// step's three load streams at 1.7 (X) + 0.75 (s | 1/s) + 1.45 (W) of the 128 -> 128 layer's 26.5 us, roughly additive with the
// quantisation (2.0) and the MFMAs (3.6): nothing overlaps much at two to three workgroups per CU, so bytes through the vector-memory
// path are worth removing even though no single stream binds.
//   * A thread stages PQ = BN / 16 CONSECUTIVE pixels of one image row. The three kw taps of a kernel row (kh) read the same PQ + 2
//     input pixels shifted by one, so ONE set of PQ + 2 loads serves three K steps (2 instead of PQ loads per step and thread), and it
//     is issued a whole kernel row — three steps — before its first use.
//   * s | 1/s of all K columns are copied to LDS once per workgroup (2 K floats, read as f32x4 quads: ds_read_b128) — no global load
//     per step. (Staging them per kernel row instead, 3 KB whatever K, measured slower twice; at Cin = 512 the resident block costs a
//     workgroup per CU, so that layer keeps the per-step kernel.)
//   * Pixel p of the tile sits in LDS row rho(p) = (p & ~3) | ((p + (p >> 2)) & 3). ds_write_b32 is served in two groups of 32 lanes
//     on 32 banks (MI355X_MICROARCH.md, LDS table): a group holds the code dwords of two pixel groups, whose rows are PQ apart — 256
//     bytes at PQ = 4, the same 16 banks. rho rotates every aligned group of four rows by its index, which puts them 64 bytes apart
//     (conflict-free at PQ = 4 by the guide's rules; PQ = 2 keeps a 2-way conflict, which a ds_write_b32 absorbs). Rows move only inside
//     aligned groups of four, which the XOR swizzle of the chunk position does not see: the B-fragment reads stay conflict-free
//     (PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 1 % / 8 % / 0 % for <4, 2> / <4, 1> / <2, 2>).
// Requires ks = 3, stride 1, pad 1 (Ho = H, Wo = W), Wo a multiple of PQ (a thread's pixels never straddle an image row).
template <int TI, int TJ>
__global__ __launch_bounds__(256, 3) void conv2d_i8r_kernel(ConvGroup grp) {
    constexpr int NTH = 256, NWN = 2;
    const ConvArgs a = grp.p[blockIdx.z];
    if ((int)blockIdx.y * 32 * TI >= a.Cout) return;
    constexpr int BM = 32 * TI, BN = 16 * TJ * NWN, PQ = BN / 16, NX = PQ + 2;
    constexpr int LD = 64;
    auto swz = [](int row, int chunk) { return chunk ^ ((0x1320 >> (((row >> 2) & 3) * 4)) & 3); };
    auto rho = [](int p) { return (p & ~3) | ((p + (p >> 2)) & 3); };
    constexpr int WPT = (BM * 4 + NTH - 1) / NTH;
    static_assert(BM >= 64 && (PQ == 2 || PQ == 4), "tile shape");
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * LD];
    __shared__ unsigned int pm[16 * BM];
    __shared__ __attribute__((aligned(16))) float ep[4][BM];
    extern __shared__ __attribute__((aligned(16))) f32x4 sr_lds[];      // s[K] | 1/s[K], as quads (indexing by quad keeps the reads ds_read_b128:
                                                                         // through a float pointer the compiler split them into ds_read2_b32 pairs,
                                                                         // which are banked mod 32 over 32 lanes — PMC: 20 % conflict cycles)
    auto Wbuf = [&](int b) -> char * { return smem + b * (BM + BN) * LD; };
    auto Xbuf = [&](int b) -> char * { return smem + b * (BM + BN) * LD + BM * LD; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int g = lane >> 4, l15 = lane & 15;
    const int L = a.Ho * a.Wo;
    const int K = a.Cin * 9;
    const int bx = a.xcd_order ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int pix0 = bx * BN;
    if (pix0 >= L) return;
#ifdef AQ3D_KO_PLACE
    const unsigned long long ko_t0 = wall_clock64();
#endif
    const int oc0 = blockIdx.y * BM;
    const int cchunks = a.Cin >> 6;
    const int nrows = 3 * cchunks;          // staging rows: (channel chunk, kh), three K steps each
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.H * a.W * a.in_ld * 4, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.Cout * K, 0x00020000);

    // staging role: thread = (channel quad cq, pixel group pq) -> pixels pix0 + PQ * pq + {0 .. PQ-1} of ONE image row
    const int cq = tid & 15, pq = tid >> 4;
    const int p0 = pix0 + PQ * pq;
    const int ho = p0 / a.Wo, wo = p0 - ho * a.Wo;
    const bool live = p0 < L;
    const int xbase = ((ho - 1) * a.W + (wo - 1)) * a.in_ld + cq * 4;      // element offset of input pixel (ho - 1, wo - 1), channel 4 cq
    const bool left_ok = live && wo > 0, right_ok = live && wo + PQ < a.W;
    unsigned hmask = 0u;                    // kernel rows inside the image
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) hmask |= (unsigned)(live && (unsigned)(ho - 1 + kh) < (unsigned)a.H) << kh;

    f32x4 xa[NX], xb[NX];                   // the input pixels of the kernel row being quantised / of the next one, in flight
    i32x4 wreg[WPT];
    int wrow_off[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int item = min(tid + NTH * i, BM * 4 - 1);
        wrow_off[i] = min(oc0 + (item >> 2), a.Cout - 1) * K + (item & 3) * 16;
    }
    int lr_kh = 0, lr_c0 = 0;               // kernel row / channel chunk of the next load_row
    auto load_row = [&](f32x4 (&xr)[NX], bool any) {
        const bool row_ok = any && ((hmask >> lr_kh) & 1u);
        const int roff = xbase + lr_kh * a.W * a.in_ld + lr_c0;
        if (++lr_kh == 3) { lr_kh = 0; lr_c0 += 64; }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const bool ok = row_ok && (j == 0 ? left_ok : (j == NX - 1 ? right_ok : true));
            const unsigned off = ok ? (unsigned)(roff + j * a.in_ld) * 4u : 0xfffffff0u;
            xr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, off, 0, 0));
        }
    };
    int lw_tap = 0, lw_c0 = 0;              // tap / channel chunk of the next load_w: K steps run chunk-major like conv2d_i8_kernel's
    auto load_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            wreg[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (unsigned)(wrow_off[i] + lw_tap * a.Cin + lw_c0), 0, 0));
        if (++lw_tap == 9) { lw_tap = 0; lw_c0 += 64; }
    };

    load_row(xb, true);                     // kernel row 0 of chunk 0
    load_w();                               // K step 0
    if (a.pieces_out)
        for (int i = tid; i < 16 * BM; i += NTH) pm[i] = 0u;
    for (int i = tid; i < BM; i += NTH) {
        const int oc = oc0 + i;
        const bool in = oc < a.Cout;
        ep[0][i] = in ? a.out_mul[oc] : 0.f;
        ep[1][i] = (in && a.bias) ? a.bias[oc] : 0.f;
        ep[2][i] = (in && a.post_scale) ? a.post_scale[oc] : 1.f;
        ep[3][i] = (in && a.post_scale) ? a.post_shift[oc] : 0.f;
    }
    const int K4 = K >> 2;
    for (int i = tid; i < K4; i += NTH) {
        sr_lds[i] = reinterpret_cast<const f32x4 *>(a.s)[i];
        sr_lds[K4 + i] = reinterpret_cast<const f32x4 *>(a.rinv)[i];
    }
    const float sx = a.scal[0];
    const bool clamp = a.scal[3] != 0.f;

    // LDS offsets of this thread's code dwords (row rho(pixel), chunk cq >> 2) and of its fragment reads
    int xst[PQ];
#pragma unroll
    for (int i = 0; i < PQ; ++i) {
        const int row = rho(PQ * pq + i);
        xst[i] = row * LD + swz(row, cq >> 2) * 16 + (cq & 3) * 4;
    }
    int bfo[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int row = rho(wn * (BN / NWN) + j * 16 + l15);
        bfo[j] = row * LD + swz(row, g) * 16;
    }
    int qs_quad = cq, qs_tap = 0;           // quad of K columns (tap * Cin + c0) / 4 + cq of the next step's scales
    const int cin4 = a.Cin >> 2;
    struct Scales { f32x4 s, r; };
    auto read_scales = [&]() {
        Scales v;
        v.s = sr_lds[qs_quad];
        v.r = sr_lds[K4 + qs_quad];
        qs_quad += cin4;
        if (++qs_tap == 9) { qs_tap = 0; qs_quad += 16 - 9 * cin4; }
        return v;
    };
    auto quant_step = [&](auto clamp_c, auto kw_c, const Scales &sc, int buf) {
        constexpr bool CL = decltype(clamp_c)::value;
        constexpr int KW = decltype(kw_c)::value;
#pragma unroll
        for (int i = 0; i < PQ; ++i) *reinterpret_cast<int *>(Xbuf(buf) + xst[i]) = quant_pack4<CL>(xa[KW + i], sc.s, sc.r, sx);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int item = tid + NTH * i;
            if (BM * 4 % NTH == 0 || item < BM * 4)
                *reinterpret_cast<i32x4 *>(Wbuf(buf) + (item >> 2) * LD + swz(item >> 2, item & 3) * 16) = wreg[i];
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
            af[i] = *reinterpret_cast<const i32x4 *>(Wbuf(buf) + (wm * (BM / 2) + i * 16 + l15) * LD + swz(l15, g) * 16);
#pragma unroll
        for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const i32x4 *>(Xbuf(buf) + bfo[j]);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
    };

    __syncthreads();                        // s | 1/s (and pm, ep) in LDS
    // Staging row R = K steps 3R .. 3R+2. Entering it, xb holds its input pixels (issued one row earlier): they move to xa and row
    // R+1's loads start (beyond the last row: out-of-range offsets — no traffic, no branch). Step q: scales of q from LDS, the MFMAs of
    // step q-1 on the other buffer, quantise(q) + W(q) into buffer q & 1, W(q+1) issued, barrier. The first row exists twice (without
    // the leading MFMA step): a conditional inside the loop makes the compiler shuttle the accumulators (see conv2d_i8_kernel).
    int q = 0;
    auto row_steps = [&](auto clamp_c, auto first_c, int R) {
#pragma unroll
        for (int j = 0; j < NX; ++j) xa[j] = xb[j];
        load_row(xb, R + 1 < nrows);
        {
            const Scales sc = read_scales();
            __builtin_amdgcn_sched_barrier(0);
            if (!decltype(first_c)::value) mma_step((q - 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            quant_step(clamp_c, std::integral_constant<int, 0>{}, sc, q & 1);
            load_w();
            __syncthreads();
            ++q;
        }
        {
            const Scales sc = read_scales();
            __builtin_amdgcn_sched_barrier(0);
            mma_step((q - 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            quant_step(clamp_c, std::integral_constant<int, 1>{}, sc, q & 1);
            load_w();
            __syncthreads();
            ++q;
        }
        {
            const Scales sc = read_scales();
            __builtin_amdgcn_sched_barrier(0);
            mma_step((q - 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            quant_step(clamp_c, std::integral_constant<int, 2>{}, sc, q & 1);
            load_w();                       // after the last step: a column past this row's K, bounded by the buffer — never stored
            __syncthreads();
            ++q;
        }
    };
    auto k_loop = [&](auto clamp_c) {
        row_steps(clamp_c, std::true_type{}, 0);
        for (int R = 1; R < nrows; ++R) row_steps(clamp_c, std::false_type{}, R);
        mma_step((q - 1) & 1);
    };
    if (clamp) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    i8_epilogue<TI, TJ, BM, NTH>(a, acc, pm, ep, pix0 + wn * (BN / NWN), oc0, wm, L);
#ifdef AQ3D_KO_PLACE
    if (tid == 0 && g_i8_place) {
        unsigned long long *o = g_i8_place + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4;
        o[0] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        o[1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        o[2] = ko_t0;
        o[3] = wall_clock64();
    }
#endif
}

// ------------------------------------------------------------------ fp32 conv / transposed conv
// WM x WN waves; wave tile (TI*16) rows x (TJ*16) pixels; K step = KS input channels of one tap (KS = 32 halves the
// barriers but measured no faster than 16, which is what is launched).
template <int WM, int WN, int TI, int TJ, int KS>
__global__ __launch_bounds__(256) void conv2d_f32_kernel(ConvGroup grp) {
    const ConvArgs a = grp.p[blockIdx.z];
    if ((int)blockIdx.y * WM * TI * 16 >= a.rows_total) return;
    constexpr int BM = WM * TI * 16;
    constexpr int BN = WN * TJ * 16;
    constexpr int LD = KS + 4;  // floats per LDS row: KS + 4 pad
    constexpr int QPR = KS / 4; // float4 quads per row and step
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LD];
    auto Wbuf = [&](int b) -> float * { return smem + b * (BM + BN) * LD; };
    auto Xbuf = [&](int b) -> float * { return smem + b * (BM + BN) * LD + BM * LD; };
    constexpr int XI = (BN * QPR + 255) / 256;  // float4 items per thread for X (BN rows x QPR quads)
    constexpr int WI = (BM * QPR + 255) / 256;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int g = lane >> 4, l15 = lane & 15;
    const int L = a.Ho * a.Wo;  // GEMM columns (for deconv Ho,Wo are the INPUT grid: one column per input pixel)
    const int K = a.Cin * a.ks * a.ks;
    const int pix0 = blockIdx.x * BN;
    const int row0 = blockIdx.y * BM;
    const int cchunks = a.Cin / KS;
    const int nk = a.ks * a.ks * cchunks;

    int ph[XI], pw[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int item = tid + 256 * i;
        const int p = pix0 + item / QPR;
        if (item / QPR < BN && p < L) {
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
        const int tap = kstep / cchunks, c0 = (kstep - tap * cchunks) * KS;
        const int kh = tap / a.ks, kw = tap - kh * a.ks;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int item = tid + 256 * i;
            const int hi = ph[i] + kh, wi = pw[i] + kw;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W)
                v = *reinterpret_cast<const f32x4 *>(a.x + ((size_t)hi * a.W + wi) * a.in_ld + c0 + (item % QPR) * 4);
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int item = tid + 256 * i;
            const int row = item / QPR;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (row < BM && row0 + row < a.rows_total)
                v = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(a.w) + (size_t)(row0 + row) * K +
                                                      (size_t)tap * a.Cin + c0 + (item % QPR) * 4);
            wr[i] = v;
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int item = tid + 256 * i;
            if (item / QPR < BN) *reinterpret_cast<f32x4 *>(Xbuf(buf) + (item / QPR) * LD + (item % QPR) * 4) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int item = tid + 256 * i;
            if (item / QPR < BM) *reinterpret_cast<f32x4 *>(Wbuf(buf) + (item / QPR) * LD + (item % QPR) * 4) = wr[i];
        }
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __shared__ unsigned int pm[16 * BM];     // piece maxima of this workgroup's outputs (bit patterns of |v|)
    if (a.pieces_out)
        for (int i = tid; i < 16 * BM; i += 256) pm[i] = 0u;      // ordered before its use by the K loop's barriers

    auto mma_step = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < KS / 16; ++kk) {
            f32x4 af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
                af[i] = *reinterpret_cast<const f32x4 *>(Wbuf(buf) + (wm * TI * 16 + i * 16 + l15) * LD + kk * 16 + g * 4);
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                bf[j] = *reinterpret_cast<const f32x4 *>(Xbuf(buf) + (wn * TJ * 16 + j * 16 + l15) * LD + kk * 16 + g * 4);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
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
                    // workgroup-level merge in LDS first (a broadcast read filters nearly every element; the global slots took one
                    // dependent L2 read + atomic per OUTPUT ELEMENT before, which was most of this kernel's run time)
                    const int OW = d > 0 ? a.Wo * d : a.Wo, OH = d > 0 ? a.Ho * d : a.Ho;
                    const int oh = (int)(opix / OW), ow = (int)(opix - (size_t)oh * OW);
                    unsigned int *slot = &pm[(piece_cls(oh, OH) * 4 + piece_cls(ow, OW)) * BM + (row - row0)];
                    const unsigned int vb = __float_as_uint(fabsf(t));
                    if (vb > *slot) atomicMax(slot, vb);
                }
            }
        }
    }
    if (a.pieces_out) {
        __syncthreads();
        for (int i = tid; i < 16 * BM; i += 256) {
            const unsigned int v = pm[i];
            const int row = row0 + i % BM;
            if (v != 0u && row < a.rows_total) {
                const int oc = d > 0 ? row % a.Cout : row;
                atomicMax(reinterpret_cast<unsigned int *>(a.pieces_out) + (size_t)(i / BM) * a.pieces_ld + a.out_coff + oc, v);
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
    a->pieces_mode = 0;
    a->xcd_order = 0;
    a->w = nullptr;
    a->rows_total = g->Cout;
    return AQ3D_OK;
}

// ---- host side: per-problem argument preparation, then one launch per group of same-class problems
static int prep_i8(ConvArgs *a, const aq3d_conv2d_problem *q) {
    const aq3d_conv2d_geom *g = q->geom;
    int rc = fill_args(a, q->x, g, q->bias, q->post_scale, q->post_shift, q->y, "aq3d_conv2d_i8");
    if (rc != AQ3D_OK) return rc;
    AQ3D_REQUIRE(!q->pieces_out || (q->pieces_ld >= g->out_coff + g->Cout && g->Ho >= 3 && g->Wo >= 3), "aq3d_conv2d_i8: bad pieces_out layout");
    a->pieces_out = q->pieces_out;
    a->pieces_ld = q->pieces_ld;
    a->pieces_mode = q->pieces_mode;
    AQ3D_REQUIRE(q->pieces_mode == 0 || q->pieces_mode == 1, "aq3d_conv2d_i8: pieces_mode=%d", q->pieces_mode);
    AQ3D_REQUIRE(q->weight_or_params, "aq3d_conv2d_i8: null params");
    AQ3D_REQUIRE((g->Cin & 63) == 0, "aq3d_conv2d_i8: Cin=%d must be a multiple of 64", g->Cin);
    AQ3D_REQUIRE((g->Cout & 3) == 0 && (g->out_ld & 3) == 0 && (g->out_coff & 3) == 0, "aq3d_conv2d_i8: Cout/out_ld/out_coff must be multiples of 4");
    AQ3D_REQUIRE(g->deconv == 0, "aq3d_conv2d_i8: deconv unsupported");
    AQ3D_REQUIRE((int64_t)g->H * g->W * g->in_ld * 4 < (int64_t)0x7fffffff && (int64_t)g->Cout * g->Cin * g->ksize * g->ksize < (int64_t)0x7fffffff,
                 "aq3d_conv2d_i8: image or weight tensor beyond the 2 GiB buffer-addressing range");
    AQ3D_REQUIRE(g->Ho == (g->H + 2 * g->pad - g->ksize) / g->stride + 1 && g->Wo == (g->W + 2 * g->pad - g->ksize) / g->stride + 1,
                 "aq3d_conv2d_i8: Ho/Wo inconsistent with geometry");
    const SqLayout l = sq_layout(g->Cin, g->Cout, g->ksize);
    const char *base = (const char *)q->weight_or_params;
    a->s = (const float *)(base + l.s);
    a->rinv = (const float *)(base + l.r);
    a->out_mul = (const float *)(base + l.mul);
    a->scal = (const float *)(base + l.scal);
    a->w = base + l.wq;
    return AQ3D_OK;
}

static bool same_class(const aq3d_conv2d_geom *a, const aq3d_conv2d_geom *b) {
    return a->H == b->H && a->W == b->W && a->Cin == b->Cin && a->ksize == b->ksize && a->stride == b->stride && a->pad == b->pad &&
           a->Ho == b->Ho && a->Wo == b->Wo && a->deconv == b->deconv;
}

extern "C" int aq3d_conv2d_i8_group(const aq3d_conv2d_problem *probs, int32_t n, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(probs && n >= 1 && n <= AQ3D_MAX_GROUP, "aq3d_conv2d_i8_group: n=%d not in 1..%d", n, AQ3D_MAX_GROUP);
    ConvGroup grp;
    int max_cout = 0;
    for (int i = 0; i < n; ++i) {
        AQ3D_REQUIRE(probs[i].geom, "aq3d_conv2d_i8_group: null geometry");
        int rc = prep_i8(&grp.p[i], &probs[i]);
        if (rc != AQ3D_OK) return rc;
        AQ3D_REQUIRE(same_class(probs[0].geom, probs[i].geom) && (probs[0].geom->Cout >= 128) == (probs[i].geom->Cout >= 128),
                     "aq3d_conv2d_i8_group: problem %d is not of the launch class of problem 0", i);
        max_cout = probs[i].geom->Cout > max_cout ? probs[i].geom->Cout : max_cout;
    }
    for (int i = n; i < AQ3D_MAX_GROUP; ++i) grp.p[i] = grp.p[0];
    const aq3d_conv2d_geom *g = probs[0].geom;
    const int L = g->Ho * g->Wo;
    // pixel-tile width: enough workgroups for >= 2 per CU (the K loop is a latency chain, it needs co-resident partners)
    const int ocb = max_cout >= 128 ? aq3d_div_up(max_cout, 128) : aq3d_div_up(max_cout, 64);
    int bn = 128;
    if (aq3d_div_up(L, 128) * ocb * n < 512) bn = 64;
    if (aq3d_div_up(L, 64) * ocb * n < 512) bn = 32;
    // where 128-pixel tiles would be picked (the head's grouped 64 -> 64 convs) the row-register kernel on 64-pixel tiles moves fewer
    // bytes per pixel (X 8 + W 4 KB per 64 pixels and step against 16 + 2 + 4 of s | 1/s): family time -2.3 %, frame latency -25 us,
    // frame rate unchanged (profiles/r04/i8_rows_group64_ab.jsonl)
    if (bn == 128 && !(AQ3D_EXP & 8) && g->ksize == 3 && g->stride == 1 && g->pad == 1 && g->Wo % 4 == 0 && g->Cin <= 256) bn = 64;
    // (also measured: the per-column scales s | 1/s staged once per workgroup in LDS instead of two global loads per thread and step —
    // 2K floats of dynamic LDS cost more occupancy than the loads: Cin = 256: 50 -> 73 us, Cin = 512: 71 -> 101 us, Cin = 128 unchanged)
    // (measured round 2, W150k: forcing 32-pixel tiles or 8 waves per workgroup on the 188 x 188 layers — more waves per SIMD — is
    // 25-30 % SLOWER, and so is a two-step-deep load prefetch: per-workgroup fixed work and weight staging, not load latency, bound it)
    // (measured round 3: 48-pixel tiles on 6 waves, <4, 1, 3>: 737 workgroups = 2.9 per CU instead of 553 = 2.2 with some CUs at 3 — or
    // 80-pixel tiles on 10 waves — 28.4 -> 33.5 / 34.7 us for 128 -> 128 at 188 x 188: four MFMAs per wave and step do not cover the step's chain)
    // pixel tiles are dealt to the 8 XCDs in contiguous ranges (grid x padded to a multiple of 8): vertically adjacent tiles of a
    // 3x3 conv re-read each other's input rows and then share an L2
    for (int i = 0; i < AQ3D_MAX_GROUP; ++i) grp.p[i].xcd_order = 1;
#define AQ3D_I8_LAUNCH(TI_, TJ_) \
    hipLaunchKernelGGL((conv2d_i8_kernel<TI_, TJ_, 2>), dim3(aq3d_div_up(aq3d_div_up(L, 32 * TJ_), 8) * 8, ocb, n), dim3(256), 0, stream, grp)
    // 3x3 / stride 1 / pad 1 with rows a thread's bn / 16 consecutive pixels cannot straddle: the row-register kernel
    // (128-pixel tiles keep the per-step kernel: ten pixels per row and thread in two register sets leave one wave per SIMD)
    const bool rows = !(AQ3D_EXP & 8) && g->ksize == 3 && g->stride == 1 && g->pad == 1 && bn <= 64 && g->Wo % (bn / 16) == 0 && g->Cin <= 256;
    const size_t sr_bytes = (size_t)2 * 9 * g->Cin * sizeof(float);
#define AQ3D_I8R_LAUNCH(TI_, TJ_) \
    hipLaunchKernelGGL((conv2d_i8r_kernel<TI_, TJ_>), dim3(aq3d_div_up(aq3d_div_up(L, 32 * TJ_), 8) * 8, ocb, n), dim3(256), sr_bytes, stream, grp)
    if (rows) {
        if (g->Cout >= 128) {
            if (bn == 64) AQ3D_I8R_LAUNCH(4, 2);
            else AQ3D_I8R_LAUNCH(4, 1);
        } else {
            if (bn == 64) AQ3D_I8R_LAUNCH(2, 2);
            else AQ3D_I8R_LAUNCH(2, 1);
        }
    } else if (g->Cout >= 128) {
        if (bn == 128) AQ3D_I8_LAUNCH(4, 4);
        else if (bn == 64) AQ3D_I8_LAUNCH(4, 2);
        else AQ3D_I8_LAUNCH(4, 1);
    } else {
        if (bn == 128) AQ3D_I8_LAUNCH(2, 4);
        else if (bn == 64) AQ3D_I8_LAUNCH(2, 2);
        else AQ3D_I8_LAUNCH(2, 1);
    }
#undef AQ3D_I8_LAUNCH
#undef AQ3D_I8R_LAUNCH
    AQ3D_CHECK_LAUNCH("aq3d_conv2d_i8");
    return AQ3D_OK;
}

extern "C" int aq3d_conv2d_i8(const float *x, const void *params, const aq3d_conv2d_geom *g, const float *bias,
                              const float *post_scale, const float *post_shift, float *y, float *pieces_out,
                              int32_t pieces_ld, int32_t pieces_mode, void *stream_) {
    AQ3D_REQUIRE(g, "aq3d_conv2d_i8: null argument");
    aq3d_conv2d_problem q;
    q.weight_split = nullptr;
    q.x = x; q.weight_or_params = params; q.geom = g; q.bias = bias; q.post_scale = post_scale; q.post_shift = post_shift;
    q.y = y; q.pieces_out = pieces_out; q.pieces_ld = pieces_ld; q.pieces_mode = pieces_mode;
    return aq3d_conv2d_i8_group(&q, 1, stream_);
}

// ------------------------------------------------------------------ fp32 conv on the fp16 matrix pipe (split operands)
// Same tiling and staging as conv2d_f32_kernel, but the products run as v_mfma_f32_16x16x32_f16: every fp32 operand x is staged in
// LDS as hi = f16(x) and lo' = f16((x - hi) * 2^11) (x == hi + lo exactly in fp32; lo' carries 11 more bits), and
//     acc  += hi_w * hi_x            accl += hi_w * lo'_x + lo'_w * hi_x            y = acc + accl * 2^-11
// with fp32 accumulation. Dropped: the lo*lo term and the rounding of lo', both ~2^-22 relative to the product — the result
// differs from exact fp32 products by ~5e-7 relative, below the reordering noise of any fp32 GEMM (and far below the TF32 the
// reference's cuDNN uses for these convolutions on Ampere). 3 MFMAs of 16 cycles per 32-deep K step instead of 8 of 32 cycles:
// the fp32 matrix pipe (157 TFLOP/s) was what bound the deblocks. |x| must stay below the fp16 range (65504); the exact
// fp32-pipe kernel above takes the layers whose Cin is not a multiple of 32.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// SQ = true: SQConv2d with more than 8 activation bits (the reference's "16-bit activation preservation", README.md:10,21,
// act_bits of quant_centerpoint.py:74-83). The codes no longer fit the int8 pipe, so the layer runs on fake-quantised VALUES,
// exactly what the reference multiplies: x' = clamp(rint((x / s) * sx)) / sx per (pixel, tap, channel) while the tile is
// staged (exact quotients by Markstein correction), weights = the int8 codes of aq3d_sq_prep as fp32 (exact in the hi half),
// per-channel weight step applied in the epilogue (a.out_mul).
// K1 = true: a plain GEMM over the pixels (1x1 conv, or kernel == stride transposed conv) with the weights already split into fp16
// planes by aq3d_conv2d_split_weight: the per-item offsets and bounds are computed once (a step only adds the channel offset as the
// buffer load's scalar offset), the weight halves go from global memory to LDS as they are. The three deblocks of the BEV backbone:
// of ~280 instructions per K step of the general loop ~100 were address / bounds / branch work and ~50 the weight split that every one
// of the 550-1100 pixel tiles repeated.
template <int WM, int WN, int TI, int TJ, int KS, bool SQ = false, bool K1 = false>
__global__ __launch_bounds__(256) void conv2d_f16s_kernel(ConvGroup grp) {
    const ConvArgs a = grp.p[blockIdx.z];
    // Tile order: the row (output-channel) blocks of ONE pixel tile re-read the same input pixels, so they are dispatched back to
    // back on the SAME XCD (workgroups go round-robin over the 8 XCDs, each with its own L2): id -> xcd = id % 8, j = id / 8,
    // row block = j % NRB, pixel tile = (j / NRB) * 8 + xcd. With (pixel tile, row block) as (blockIdx.x, blockIdx.y) the 4-16 row
    // blocks of a tile ran a whole grid row apart on arbitrary XCDs and each fetched its input again from beyond L2
    // (PMC, round 2: 130 MB per launch for 21 MB of algorithmic traffic). gridDim.x = 8 * ceil(pixel tiles / 8) * NRB.
    const int nrb = a.nrb;
    const int bid = (int)blockIdx.x;
    const int bj = bid >> 3;
    const int rb = bj % nrb, pt = (bj / nrb) * 8 + (bid & 7);
    if (rb * WM * TI * 16 >= a.rows_total) return;
    constexpr int BM = WM * TI * 16;
    constexpr int BN = WN * TJ * 16;
    constexpr int LD = 2 * KS + 8;  // halves per LDS row: KS hi | KS lo | 16 B pad
    constexpr int QPR = KS / 4;     // float4 quads per row and step
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * (BM + BN) * LD];
    auto Wbuf = [&](int b) -> _Float16 * { return smem + b * (BM + BN) * LD; };
    auto Xbuf = [&](int b) -> _Float16 * { return smem + b * (BM + BN) * LD + BM * LD; };
    constexpr int XI = (BN * QPR + 255) / 256;
    constexpr int WI = (BM * QPR + 255) / 256;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int g = lane >> 4, l15 = lane & 15;
    const int L = a.Ho * a.Wo;
    const int K = a.Cin * a.ks * a.ks;
    const int pix0 = pt * BN;
    const int row0 = rb * BM;
    if (pix0 >= a.Ho * a.Wo) return;                     // padding of the pixel-tile count to a multiple of 8
    const int cchunks = a.Cin / KS;
    const int nk = a.ks * a.ks * cchunks;

    int ph[XI], pw[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int item = tid + 256 * i;
        const int p = pix0 + item / QPR;
        if (item / QPR < BN && p < L) {
            int ho = p / a.Wo;
            ph[i] = ho * a.stride - a.pad;
            pw[i] = (p - ho * a.Wo) * a.stride - a.pad;
        } else {
            ph[i] = -0x10000;
            pw[i] = -0x10000;
        }
    }
    f32x4 xr[XI], wr[WI];
    f32x4 sqs = (f32x4){1.f, 1.f, 1.f, 1.f}, sqr = sqs;       // SQ: smoothing scale and its reciprocal of this thread's 4 columns
    const float sq_sx = SQ ? a.scal[0] : 0.f, sq_rsx = SQ ? a.scal[2] : 0.f, sq_qmax = SQ ? a.scal[4] : 0.f;
    auto load_step = [&](int kstep) {
        const int tap = kstep / cchunks, c0 = (kstep - tap * cchunks) * KS;
        const int kh = tap / a.ks, kw = tap - kh * a.ks;
        if (SQ) {       // 256 % QPR == 0: the column quad of a thread is the same for all its items
            sqs = *reinterpret_cast<const f32x4 *>(a.s + (size_t)tap * a.Cin + c0 + (tid % QPR) * 4);
            sqr = *reinterpret_cast<const f32x4 *>(a.rinv + (size_t)tap * a.Cin + c0 + (tid % QPR) * 4);
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int item = tid + 256 * i;
            const int hi = ph[i] + kh, wi = pw[i] + kw;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W)
                v = *reinterpret_cast<const f32x4 *>(a.x + ((size_t)hi * a.W + wi) * a.in_ld + c0 + (item % QPR) * 4);
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int item = tid + 256 * i;
            const int row = item / QPR;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (row < BM && row0 + row < a.rows_total)
                v = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(a.w) + (size_t)(row0 + row) * K +
                                                      (size_t)tap * a.Cin + c0 + (item % QPR) * 4);
            wr[i] = v;
        }
    };
    auto put_split = [&](_Float16 *row, int q, f32x4 v) {
        f16x4 h, l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            h[r] = (_Float16)v[r];
            l[r] = (_Float16)__fmul_rn(__fsub_rn(v[r], (float)h[r]), 2048.f);
        }
        *reinterpret_cast<f16x4 *>(row + q * 4) = h;
        *reinterpret_cast<f16x4 *>(row + KS + q * 4) = l;
    };
    auto store_step = [&](int buf) {
        if (SQ) {       // TensorQuantizer on X / s: q = clamp(rint(t * sx), -qmax - 1, qmax), value q / sx (sx = 0 / divide by 1 when amax <= 2^-24)
#pragma unroll
            for (int i = 0; i < XI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float q = rintf(__fmul_rn(div_by_scale(xr[i][r], sqs[r], sqr[r]), sq_sx));
                    q = fminf(fmaxf(q, -sq_qmax - 1.f), sq_qmax);
                    xr[i][r] = sq_sx != 0.f ? div_by_scale(q, sq_sx, sq_rsx) : q;
                }
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int item = tid + 256 * i;
            if (item / QPR < BN) put_split(Xbuf(buf) + (item / QPR) * LD, item % QPR, xr[i]);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int item = tid + 256 * i;
            if (item / QPR < BM) put_split(Wbuf(buf) + (item / QPR) * LD, item % QPR, wr[i]);
        }
    };

    // ---- K1: GEMM staging
    constexpr int WQ = KS / 8;                          // 16-byte groups of 8 halves per row, plane and step
    constexpr int WI16 = (BM * WQ + 255) / 256;
    unsigned xoff[XI], woff[WI16];
    i32x4 wh[WI16], wl[WI16];
    const auto rsrc_x1 = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, K1 ? a.H * a.W * a.in_ld * 4 : 0, 0x00020000);
    const auto rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc((void *)a.w16, 0, K1 ? a.rows_total * K * 4 : 0, 0x00020000);
    if (K1) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int item = tid + 256 * i, pix = item / QPR, p = pix0 + pix;
            xoff[i] = (pix < BN && p < L) ? (unsigned)(p * a.in_ld + (item % QPR) * 4) * 4u : 0xfffffff0u;
        }
#pragma unroll
        for (int i = 0; i < WI16; ++i) {
            const int item = tid + 256 * i, row = item / WQ;
            woff[i] = (row < BM && row0 + row < a.rows_total) ? (unsigned)((row0 + row) * 2 * K + (item % WQ) * 8) * 2u : 0xfffffff0u;
        }
    }
    auto load_step1 = [&](int kstep) {       // out-of-range items: the vector offset alone is range-checked, they read zeros
        const int c0 = kstep * KS;
#pragma unroll
        for (int i = 0; i < XI; ++i) xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x1, xoff[i], c0 * 4, 0));
#pragma unroll
        for (int i = 0; i < WI16; ++i) {
            wh[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w1, woff[i], c0 * 2, 0));
            wl[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w1, woff[i], (K + c0) * 2, 0));
        }
    };
    auto store_step1 = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int item = tid + 256 * i;
            if (item / QPR < BN) put_split(Xbuf(buf) + (item / QPR) * LD, item % QPR, xr[i]);
        }
#pragma unroll
        for (int i = 0; i < WI16; ++i) {
            const int item = tid + 256 * i, row = item / WQ;
            if (BM * WQ % 256 == 0 || row < BM) {
                *reinterpret_cast<i32x4 *>(Wbuf(buf) + row * LD + (item % WQ) * 8) = wh[i];
                *reinterpret_cast<i32x4 *>(Wbuf(buf) + row * LD + KS + (item % WQ) * 8) = wl[i];
            }
        }
    };

    f32x4 acc[TI][TJ], accl[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accl[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    __shared__ unsigned int pm[16 * BM];     // piece maxima of this workgroup's outputs (bit patterns of |v|)
    if (a.pieces_out)
        for (int i = tid; i < 16 * BM; i += 256) pm[i] = 0u;      // ordered before its use by the K loop's barriers

    auto mma_step = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < KS / 32; ++kk) {
            f16x8 ah[TI], al[TI], bh[TJ], bl[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const _Float16 *r = Wbuf(buf) + (wm * TI * 16 + i * 16 + l15) * LD + kk * 32 + g * 8;
                ah[i] = *reinterpret_cast<const f16x8 *>(r);
                al[i] = *reinterpret_cast<const f16x8 *>(r + KS);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const _Float16 *r = Xbuf(buf) + (wn * TJ * 16 + j * 16 + l15) * LD + kk * 32 + g * 8;
                bh[j] = *reinterpret_cast<const f16x8 *>(r);
                bl[j] = *reinterpret_cast<const f16x8 *>(r + KS);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    accl[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], accl[i][j], 0, 0, 0);
                    accl[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], accl[i][j], 0, 0, 0);
                }
        }
    };
    // epilogue operands of the lane's row quads, fetched before the K loop (it has 4-8 steps: a global round trip at the tail shows)
    const bool vec4 = ((a.Cout | a.out_ld | a.out_coff | a.rows_total) & 3) == 0;      // wave-uniform
    f32x4 e_b[TI], e_ps[TI], e_pt[TI], e_om[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        e_b[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; e_pt[i] = e_b[i];
        e_ps[i] = (f32x4){1.f, 1.f, 1.f, 1.f}; e_om[i] = e_ps[i];
        const int rbase = row0 + wm * TI * 16 + i * 16 + 4 * g;
        if (vec4 && rbase < a.rows_total) {
            const int oc = a.deconv > 0 ? rbase % a.Cout : rbase;
            if (SQ) e_om[i] = *reinterpret_cast<const f32x4 *>(a.out_mul + oc);
            if (a.bias) e_b[i] = *reinterpret_cast<const f32x4 *>(a.bias + oc);
            if (a.post_scale) {
                e_ps[i] = *reinterpret_cast<const f32x4 *>(a.post_scale + oc);
                e_pt[i] = *reinterpret_cast<const f32x4 *>(a.post_shift + oc);
            }
        }
    }
    if (K1) {
        load_step1(0);
        store_step1(0);
        __syncthreads();
        for (int k = 0; k + 1 < nk; ++k) {
            const int buf = k & 1;
            load_step1(k + 1);
            mma_step(buf);
            store_step1(buf ^ 1);
            __syncthreads();
        }
    } else {
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
    }
    mma_step((nk - 1) & 1);

    // epilogue: a lane holds rows rbase .. rbase+3 of pixel column l15. With 4-aligned channel geometry the four rows are four
    // consecutive output channels of ONE output pixel (deconv: row = (i*d + j) * Cout + oc, Cout % 4 == 0): one 16-byte store
    // instead of four scalar ones, bias / BN loaded as vectors. (The deblocks write 36 MB each; the K loop has 4-8 steps, so the
    // epilogue is a large part of the kernel.)
    const int d = a.deconv;
    const int OW = d > 0 ? a.Wo * d : a.Wo, OH = d > 0 ? a.Ho * d : a.Ho;
    f32x4 vout[TI][TJ];                 // vec4 path: values kept in registers, stored AFTER the piece maxima are flushed
    size_t vdst[TI][TJ];
    int ph_[TJ], pw_[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int p = pix0 + wn * TJ * 16 + j * 16 + l15;
        ph_[j] = p < L ? p / a.Wo : -1;
        pw_[j] = p - ph_[j] * a.Wo;
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int rbase = row0 + wm * TI * 16 + i * 16 + 4 * g;
        if (vec4) {
            if (rbase >= a.rows_total) continue;
            const int ij = d > 0 ? rbase / a.Cout : 0;
            const int oc = rbase - ij * a.Cout;
            const int di = d > 0 ? ij / d : 0, dj = d > 0 ? ij - di * d : 0;
            const f32x4 b = e_b[i], ps = e_ps[i], pt = e_pt[i], om = e_om[i];
            float mint[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                if (ph_[j] < 0) continue;
                const int oh = d > 0 ? ph_[j] * d + di : ph_[j], ow = d > 0 ? pw_[j] * d + dj : pw_[j];
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = __fadd_rn(acc[i][j][r], __fmul_rn(accl[i][j][r], 1.0f / 2048.f));
                    if (SQ) t = __fmul_rn(t, om[r]);
                    if (a.bias) t = __fadd_rn(t, b[r]);
                    if (a.post_scale) t = __fadd_rn(__fmul_rn(t, ps[r]), pt[r]);
                    if (a.relu) t = fmaxf(t, 0.f);
                    v[r] = t;
                }
                vout[i][j] = v;
                vdst[i][j] = ((size_t)oh * OW + ow) * a.out_ld + a.out_coff + oc;
                if (a.pieces_out) {
                    if (a.pieces_mode == 1 && oh > 0 && oh < OH - 1 && ow > 0 && ow < OW - 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) mint[r] = fmaxf(mint[r], fabsf(v[r]));   // interior: reduced in registers below
                    } else {
                        const int pc = piece_cls(oh, OH) * 4 + piece_cls(ow, OW);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            unsigned int *slot = &pm[pc * BM + (rbase - row0) + r];
                            const unsigned int vb = __float_as_uint(fabsf(v[r]));
                            if (vb > *slot) atomicMax(slot, vb);
                        }
                    }
                }
            }
            if (a.pieces_out && a.pieces_mode == 1) {
                // merged interior class (see conv2d_i8_kernel): max over the wave's 16 pixel lanes in registers, one LDS atomic per row
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float m = mint[r];      // DPP row reduction over the 16 pixel lanes (values >= 0), result in the row's last lane
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x111, 0xf, 0xf, true)));
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x112, 0xf, 0xf, true)));
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x114, 0xf, 0xf, true)));
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x118, 0xf, 0xf, true)));
                    if (l15 == 15 && m > 0.f) atomicMax(&pm[10 * BM + (rbase - row0) + r], __float_as_uint(m));
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            if (ph_[j] < 0) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rbase + r;
                if (row >= a.rows_total) continue;
                int oc = row, oh = ph_[j], ow = pw_[j];
                if (d > 0) {  // row = (i*d + j)*Cout + oc ; input pixel (h, w) -> output (h*d+i, w*d+j)
                    const int ij = row / a.Cout;
                    oc = row - ij * a.Cout;
                    oh = ph_[j] * d + ij / d;
                    ow = pw_[j] * d + ij % d;
                }
                float t = __fadd_rn(acc[i][j][r], __fmul_rn(accl[i][j][r], 1.0f / 2048.f));
                if (SQ) t = __fmul_rn(t, a.out_mul[oc]);
                if (a.bias) t = __fadd_rn(t, a.bias[oc]);
                if (a.post_scale) t = __fadd_rn(__fmul_rn(t, a.post_scale[oc]), a.post_shift[oc]);
                if (a.relu) t = fmaxf(t, 0.f);
                a.y[((size_t)oh * OW + ow) * a.out_ld + a.out_coff + oc] = t;
                if (a.pieces_out) {
                    unsigned int *slot = &pm[(piece_cls(oh, OH) * 4 + piece_cls(ow, OW)) * BM + (row - row0)];
                    const unsigned int vb = __float_as_uint(fabsf(t));
                    if (vb > *slot) atomicMax(slot, vb);
                }
            }
        }
    }
    if (a.pieces_out) {
        __syncthreads();
        for (int i = tid; i < 16 * BM; i += 256) {
            const unsigned int v = pm[i];
            const int row = row0 + i % BM;
            if (v != 0u && row < a.rows_total) {
                const int oc = d > 0 ? row % a.Cout : row;
                int pc = i / BM;
                if (a.pieces_mode == 1 && pc == 10) {       // spread the merged interior maximum over the four interior slots (L2 atomics on one address serialise)
                    const int alt[4] = {10, 11, 14, 15};
                    pc = alt[blockIdx.x & 3];
                }
                atomicMax(reinterpret_cast<unsigned int *>(a.pieces_out) + (size_t)pc * a.pieces_ld + a.out_coff + oc, v);
            }
        }
    }
    if (vec4) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (row0 + wm * TI * 16 + i * 16 + 4 * g >= a.rows_total) continue;
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                if (ph_[j] >= 0) *reinterpret_cast<f32x4 *>(a.y + vdst[i][j]) = vout[i][j];
        }
    }
}

static int prep_f32(ConvArgs *a, const aq3d_conv2d_problem *q, int *L_out) {
    const aq3d_conv2d_geom *g = q->geom;
    int rc = fill_args(a, q->x, g, q->bias, q->post_scale, q->post_shift, q->y, "aq3d_conv2d_f32");
    if (rc != AQ3D_OK) return rc;
    AQ3D_REQUIRE(!q->pieces_out || q->pieces_ld >= g->out_coff + g->Cout, "aq3d_conv2d_f32: bad pieces_out layout");
    a->pieces_out = q->pieces_out;
    a->pieces_ld = q->pieces_ld;
    a->pieces_mode = q->pieces_out ? q->pieces_mode : 0;
    AQ3D_REQUIRE(a->pieces_mode == 0 || a->pieces_mode == 1, "aq3d_conv2d_f32: pieces_mode=%d", q->pieces_mode);
    AQ3D_REQUIRE(a->pieces_mode == 0 || ((g->Cout | g->out_ld | g->out_coff) & 3) == 0, "aq3d_conv2d_f32: pieces_mode 1 needs 4-aligned channel geometry");
    AQ3D_REQUIRE(q->weight_or_params, "aq3d_conv2d_f32: null weight");
    AQ3D_REQUIRE((g->Cin & 15) == 0, "aq3d_conv2d_f32: Cin=%d must be a multiple of 16", g->Cin);
    a->w = q->weight_or_params;
    a->w16 = q->weight_split;
    if (g->deconv > 0) {
        AQ3D_REQUIRE(g->ksize == 1 && g->stride == 1 && g->pad == 0, "aq3d_conv2d_f32: deconv needs ksize=1 stride=1 pad=0 (kernel == stride == deconv)");
        AQ3D_REQUIRE(g->Ho == g->H && g->Wo == g->W, "aq3d_conv2d_f32: deconv takes Ho=H, Wo=W (the input grid)");
        a->rows_total = g->deconv * g->deconv * g->Cout;
        *L_out = g->H * g->W;
    } else {
        AQ3D_REQUIRE(g->Ho == (g->H + 2 * g->pad - g->ksize) / g->stride + 1 && g->Wo == (g->W + 2 * g->pad - g->ksize) / g->stride + 1,
                     "aq3d_conv2d_f32: Ho/Wo inconsistent with geometry");
        *L_out = g->Ho * g->Wo;
    }
    return AQ3D_OK;
}

static int f32_class(int rows, int L) { return rows <= 16 ? 0 : ((rows <= 64 || L < 128 * 150) ? 1 : 2); }

// ------------------------------------------------------------------ fp32 3x3 conv with 1-4 output channels: plain VALU
// The last conv of every CenterHead branch (64 -> 1..3 channels, left in fp32 by the recipe: quant_centerpoint.py:28-71) has so few
// output channels that a GEMM tile is almost all padding: on the matrix pipe (16-row tiles, every fp32 input split into fp16 halves
// once per tap) the five branches of the Waymo head cost 72 us — all of it staging. Here a thread owns one output pixel and keeps its
// <= 4 sums in registers; the input goes through LDS once per 16-channel chunk as an 18 x 18 halo tile (each element fetched once per
// tile, not once per tap), the weights of a (tap, channel) are wave-uniform scalar operands of the fp32 FMAs. Sum order: channel chunk,
// tap, channel — a fixed fp32 FMA chain (the oracle's conv sums in another order: same 2e-5 bar as every fp32 layer).
template <int CO>
__device__ __forceinline__ void conv2d_f32_small_body(const ConvArgs &a, float *tile) {
    constexpr int TS = 16, HS = TS + 2, CC = 16, PS = CC + 4;       // 16 x 16 outputs, 18 x 18 halo, 16-channel chunks, 80-byte pixel rows
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int tiles_x = (a.Wo + TS - 1) / TS;
    const int by = (int)blockIdx.x / tiles_x, bx = (int)blockIdx.x - by * tiles_x;
    const int oy = by * TS + ty, ox = bx * TS + tx;
    const int K = a.Cin * 9;
    const float *w = static_cast<const float *>(a.w);
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    constexpr int NL = (HS * HS * (CC / 4) + 255) / 256;             // float4 loads per thread and chunk
    f32x4 v[NL];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int item = tid + 256 * i, px = item >> 2, q = item & 3;
            const int hy = px / HS, hx = px - hy * HS;
            const int iy = by * TS + hy - 1, ix = bx * TS + hx - 1;
            v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (px < HS * HS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v[i] = *reinterpret_cast<const f32x4 *>(a.x + ((size_t)iy * a.W + ix) * a.in_ld + c0 + q * 4);
        }
    };
    auto store_chunk = [&](float *buf) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int item = tid + 256 * i, px = item >> 2, q = item & 3;
            if (px < HS * HS) *reinterpret_cast<f32x4 *>(buf + px * PS + q * 4) = v[i];
        }
    };
    const int nchunks = a.Cin / CC;
    load_chunk(0);
    store_chunk(tile);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        float *cur = tile + (c & 1) * (HS * HS * PS);
        if (c + 1 < nchunks) load_chunk((c + 1) * CC);
        // one tap at a time (rolled): its CO x 16 weights are scalar operands — unrolled over the taps the compiler hoists all 9 x CO x 16
        // scalar loads of a chunk and spills SGPRs through v_writelane
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const float *pxl = cur + ((ty + dy) * HS + tx + dx) * PS;
            f32x4 x4[CC / 4];
#pragma unroll
            for (int q = 0; q < CC / 4; ++q) x4[q] = *reinterpret_cast<const f32x4 *>(pxl + q * 4);
            const float *wt = w + (size_t)tap * a.Cin + c * CC;           // wave-uniform
#pragma unroll
            for (int o = 0; o < CO; ++o)
#pragma unroll
                for (int j = 0; j < CC; ++j) acc[o] = __fmaf_rn(x4[j >> 2][j & 3], wt[(size_t)o * K + j], acc[o]);
        }
        if (c + 1 < nchunks) store_chunk(tile + ((c + 1) & 1) * (HS * HS * PS));
        __syncthreads();
    }
    if (oy < a.Ho && ox < a.Wo) {
        float *dst = a.y + ((size_t)oy * a.Wo + ox) * a.out_ld + a.out_coff;
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            float t = acc[o];
            if (a.bias) t = __fadd_rn(t, a.bias[o]);
            if (a.post_scale) t = __fadd_rn(__fmul_rn(t, a.post_scale[o]), a.post_shift[o]);
            if (a.relu) t = fmaxf(t, 0.f);
            dst[o] = t;
        }
    }
}
__global__ __launch_bounds__(256) void conv2d_f32_small_kernel(ConvGroup grp) {
    const ConvArgs a = grp.p[blockIdx.z];
    __shared__ __attribute__((aligned(16))) float tile[2 * 18 * 18 * 20];
    switch (a.Cout) {       // uniform per workgroup; the body is unrolled over the output channels
    case 1: conv2d_f32_small_body<1>(a, tile); break;
    case 2: conv2d_f32_small_body<2>(a, tile); break;
    case 3: conv2d_f32_small_body<3>(a, tile); break;
    default: conv2d_f32_small_body<4>(a, tile); break;
    }
}

// fp32 weights [rows][K] -> fp16 planes [rows][2][K]: hi = f16(w), lo' = f16((w - hi) * 2^11) — the operand form of conv2d_f16s_kernel
__global__ __launch_bounds__(256) void split_weight_kernel(const float *__restrict__ w, long long n, int K, _Float16 *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / K;
        const int k = (int)(i - r * K);
        const float v = w[i];
        const _Float16 h = (_Float16)v;
        out[(r * 2) * K + k] = h;
        out[(r * 2 + 1) * K + k] = (_Float16)__fmul_rn(__fsub_rn(v, (float)h), 2048.f);
    }
}
extern "C" int aq3d_conv2d_split_weight(const float *weight, int64_t rows, int32_t K, void *out, void *stream_) {
    AQ3D_REQUIRE(weight && out && rows >= 1 && K >= 1, "aq3d_conv2d_split_weight: bad argument");
    hipLaunchKernelGGL(split_weight_kernel, dim3(aq3d_grid(rows * K, 256)), dim3(256), 0, (hipStream_t)stream_, weight, (long long)rows * K, K,
                       static_cast<_Float16 *>(out));
    AQ3D_CHECK_LAUNCH("aq3d_conv2d_split_weight");
    return AQ3D_OK;
}

extern "C" int aq3d_conv2d_f32_group(const aq3d_conv2d_problem *probs, int32_t n, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(probs && n >= 1 && n <= AQ3D_MAX_GROUP, "aq3d_conv2d_f32_group: n=%d not in 1..%d", n, AQ3D_MAX_GROUP);
    ConvGroup grp;
    int L = 0, max_rows = 0, cls = 0;
    for (int i = 0; i < n; ++i) {
        AQ3D_REQUIRE(probs[i].geom, "aq3d_conv2d_f32_group: null geometry");
        int Li = 0;
        int rc = prep_f32(&grp.p[i], &probs[i], &Li);
        if (rc != AQ3D_OK) return rc;
        const int ci = f32_class(grp.p[i].rows_total, Li);
        if (i == 0) { L = Li; cls = ci; }
        AQ3D_REQUIRE(same_class(probs[0].geom, probs[i].geom) && ci == cls,
                     "aq3d_conv2d_f32_group: problem %d is not of the launch class of problem 0", i);
        max_rows = grp.p[i].rows_total > max_rows ? grp.p[i].rows_total : max_rows;
    }
    for (int i = n; i < AQ3D_MAX_GROUP; ++i) grp.p[i] = grp.p[0];
    bool split_ok = true;
    for (int i = 0; i < n; ++i) split_ok = split_ok && (probs[i].geom->Cin & 31) == 0;
    bool small_ok = cls == 0;       // 3x3, stride 1, pad 1, <= 4 output channels, no piece maxima: the VALU kernel
    for (int i = 0; i < n; ++i) {
        const aq3d_conv2d_geom *g_ = probs[i].geom;
        small_ok = small_ok && g_->ksize == 3 && g_->stride == 1 && g_->pad == 1 && g_->deconv == 0 && g_->Cout >= 1 && g_->Cout <= 4 &&
                   (g_->Cin & 15) == 0 && !probs[i].pieces_out;
    }
    if (small_ok) {
        const aq3d_conv2d_geom *g_ = probs[0].geom;
        hipLaunchKernelGGL(conv2d_f32_small_kernel, dim3(aq3d_div_up(g_->Ho, 16) * aq3d_div_up(g_->Wo, 16), 1, n), dim3(256), 0, stream, grp);
        AQ3D_CHECK_LAUNCH("aq3d_conv2d_f32");
        return AQ3D_OK;
    }
    if (split_ok) {     // fp16 matrix pipe, split operands (see conv2d_f16s_kernel)
        if (cls == 0) {
            for (int i = 0; i < AQ3D_MAX_GROUP; ++i) grp.p[i].nrb = 1;
            hipLaunchKernelGGL((conv2d_f16s_kernel<1, 4, 1, 2, 32>), dim3(aq3d_div_up(aq3d_div_up(L, 128), 8) * 8, 1, n), dim3(256), 0, stream, grp);
        } else {        // 64 x 64 tiles for both deblock shapes (128-row tiles measured slower: conversion work, 63 KB of LDS; 64-deep K steps: no gain)
            const int nrb = aq3d_div_up(max_rows, 64);
            for (int i = 0; i < AQ3D_MAX_GROUP; ++i) grp.p[i].nrb = nrb;
            bool k1 = true;         // plain GEMMs with pre-split weights (the deblocks)
            for (int i = 0; i < n; ++i)
                k1 = k1 && probs[i].weight_split && probs[i].geom->ksize == 1 && probs[i].geom->stride == 1 && probs[i].geom->pad == 0 &&
                     (long long)grp.p[i].rows_total * probs[i].geom->Cin * 4 < 0x7fffffffLL &&
                     (long long)probs[i].geom->H * probs[i].geom->W * probs[i].geom->in_ld * 4 < 0x7fffffffLL;
            if (k1) hipLaunchKernelGGL((conv2d_f16s_kernel<2, 2, 2, 2, 32, false, true>), dim3(aq3d_div_up(aq3d_div_up(L, 64), 8) * 8 * nrb, 1, n), dim3(256), 0, stream, grp);
            else hipLaunchKernelGGL((conv2d_f16s_kernel<2, 2, 2, 2, 32>), dim3(aq3d_div_up(aq3d_div_up(L, 64), 8) * 8 * nrb, 1, n), dim3(256), 0, stream, grp);
        }
        AQ3D_CHECK_LAUNCH("aq3d_conv2d_f32");
        return AQ3D_OK;
    }
    if (cls == 0) {
        hipLaunchKernelGGL((conv2d_f32_kernel<1, 4, 1, 2, 16>), dim3(aq3d_div_up(L, 128), 1, n), dim3(256), 0, stream, grp);
    } else if (cls == 1) {
        hipLaunchKernelGGL((conv2d_f32_kernel<2, 2, 2, 2, 16>), dim3(aq3d_div_up(L, 64), aq3d_div_up(max_rows, 64), n), dim3(256), 0, stream, grp);
    } else {
        hipLaunchKernelGGL((conv2d_f32_kernel<2, 2, 4, 2, 16>), dim3(aq3d_div_up(L, 64), aq3d_div_up(max_rows, 128), n), dim3(256), 0, stream, grp);
    }
    AQ3D_CHECK_LAUNCH("aq3d_conv2d_f32");
    return AQ3D_OK;
}

// codes of aq3d_sq_prep -> fp32 (exact), weight step 1 / scale_w per output channel (scale_w = qmax_w / amax_w, 1 where amax_w <= 2^-24)
__global__ __launch_bounds__(256) void sq16_expand_kernel(const int8_t *__restrict__ wq, const float *__restrict__ amax_w, int Cout, int K,
                                                          float qmax_w, float *__restrict__ wf, float *__restrict__ wstep) {
    const long long total = (long long)Cout * K;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) wf[t] = (float)wq[t];
    if (blockIdx.x == 0)
        for (int o = threadIdx.x; o < Cout; o += 256) {
            const float am = amax_w[o];
            wstep[o] = am <= 1.0f / (float)(1 << 24) ? 1.f : __fdiv_rn(1.0f, __fdiv_rn(qmax_w, am));
        }
}

extern "C" size_t aq3d_conv2d_sq16_workspace_bytes(int32_t Cin, int32_t Cout, int32_t ksize) {
    return sq_align((size_t)Cout * Cin * ksize * ksize * 4) + sq_align((size_t)Cout * 4);
}

extern "C" int aq3d_conv2d_sq16(const float *x, const void *params, const aq3d_conv2d_geom *g, int32_t w_bits, int32_t a_bits,
                                const float *bias, const float *post_scale, const float *post_shift, float *y, void *workspace,
                                size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(x && params && g && y && workspace, "aq3d_conv2d_sq16: null argument");
    AQ3D_REQUIRE(a_bits > 8 && a_bits <= 16 && w_bits >= 2 && w_bits <= 8, "aq3d_conv2d_sq16: bits w=%d a=%d (8-bit activations run on aq3d_conv2d_i8)", w_bits, a_bits);
    AQ3D_REQUIRE((g->Cin & 31) == 0 && g->deconv == 0, "aq3d_conv2d_sq16: Cin=%d must be a multiple of 32, no deconv", g->Cin);
    if (workspace_bytes < aq3d_conv2d_sq16_workspace_bytes(g->Cin, g->Cout, g->ksize)) {
        aq3d_set_error("aq3d_conv2d_sq16: workspace too small");
        return AQ3D_ERR_WORKSPACE;
    }
    const int K = g->Cin * g->ksize * g->ksize;
    const SqLayout l = sq_layout(g->Cin, g->Cout, g->ksize);
    const char *base = (const char *)params;
    float *wf = (float *)workspace;
    float *wstep = (float *)((char *)workspace + sq_align((size_t)g->Cout * K * 4));
    hipLaunchKernelGGL(sq16_expand_kernel, dim3(aq3d_grid((int64_t)g->Cout * K, 256, 1024)), dim3(256), 0, stream, (const int8_t *)(base + l.wq),
                       (const float *)(base + l.amaxw), g->Cout, K, (float)((1 << (w_bits - 1)) - 1), wf, wstep);
    aq3d_conv2d_problem q;
    q.weight_split = nullptr;
    q.x = x; q.weight_or_params = wf; q.geom = g; q.bias = bias; q.post_scale = post_scale; q.post_shift = post_shift;
    q.y = y; q.pieces_out = nullptr; q.pieces_ld = 0; q.pieces_mode = 0;
    ConvGroup grp;
    int L = 0;
    int rc = prep_f32(&grp.p[0], &q, &L);
    if (rc != AQ3D_OK) return rc;
    grp.p[0].s = (const float *)(base + l.s); grp.p[0].rinv = (const float *)(base + l.r); grp.p[0].scal = (const float *)(base + l.scal);
    grp.p[0].out_mul = wstep;
    grp.p[0].nrb = aq3d_div_up(g->Cout, 64);
    for (int i = 1; i < AQ3D_MAX_GROUP; ++i) grp.p[i] = grp.p[0];
    hipLaunchKernelGGL((conv2d_f16s_kernel<2, 2, 2, 2, 32, true>), dim3(aq3d_div_up(aq3d_div_up(L, 64), 8) * 8 * grp.p[0].nrb, 1, 1), dim3(256), 0, stream, grp);
    AQ3D_CHECK_LAUNCH("aq3d_conv2d_sq16");
    return AQ3D_OK;
}

extern "C" int aq3d_conv2d_f32(const float *x, const float *weight, const aq3d_conv2d_geom *g, const float *bias,
                               const float *post_scale, const float *post_shift, float *y, float *pieces_out,
                               int32_t pieces_ld, void *stream_) {
    AQ3D_REQUIRE(g, "aq3d_conv2d_f32: null argument");
    aq3d_conv2d_problem q;
    q.weight_split = nullptr;
    q.x = x; q.weight_or_params = weight; q.geom = g; q.bias = bias; q.post_scale = post_scale; q.post_shift = post_shift;
    q.y = y; q.pieces_out = pieces_out; q.pieces_ld = pieces_ld; q.pieces_mode = 0;
    return aq3d_conv2d_f32_group(&q, 1, stream_);
}
