// Sparse conv of the QConvNd layers (quant/quant.py:36-58 around spconv) on the int8 matrix pipe: int8-CODED activations
// (1 byte per gathered element) x fixed-point weight PLANES that carry the per-input-channel activation scale.
//
// QConvNd quantises activations PER INPUT CHANNEL, so the scale d_c sits inside the K contraction:
//     y[m,o] = sum_{k,c} (q[nbr(m,k),c] * d_c) * (wq[k,o,c] * t_o)
// and a plain int8 x int8 dot cannot carry it. The gather moves the CODE q (int8) and the scale moves into the weights:
//     W'[k,o,c] = wq[k,o,c] * (d_c / S),  S = max_c d_c,  |W'| <= 127,      y = (sum q * W') * (S * t_o).
// W' is held in FIXED POINT with 16 fractional bits, W'' = rint(W' * 2^16), split into three balanced base-256 digits
// d2 d1 d0 (int8 each): q * d_p is an exact int8 x int8 product of v_mfma_i32_16x16x64_i8, every plane has its own int32
// accumulator, and the epilogue recombines them in fp32: |W' - W''/2^16| <= 2^-17, i.e. fp32-grade sums (measured <= 4e-7 of
// the output scale per layer). Rounds 1-2 multiplied fp16 codes by an fp16 hi|lo weight pair on the fp16 pipe: two MFMAs per
// product at half the int8 rate (4 int8-MFMA times against 3 here), 4 bytes of weights through LDS per product against 3,
// and a 5-instruction int8 -> fp16 conversion of every gathered dword that is gone altogether.
// AQ3D_FLAG_W16: planes d2 d1 only (balanced digits: dropping d0 ROUNDS W'' to 8 fractional bits, |error| <= 2^-9 per
// weight — 16-bit fixed-point weights, ~5e-5 of the output scale per layer): two MFMAs and two bytes per product. This is the
// north star's "int8 weights x activations with the scale folded into the gather" at the int8 pipe's rate.
//
//   sp_prep_kernel      one launch per layer: blocks [0, row_blocks) quantise the producer's fp32 rows to codes, the rest
//                       build the weight planes and the epilogue multipliers S * t_o / 2^16. Both depend on the abs-max only.
//   spconv_cd_kernel    Cin = 16, kvol = 27: all planes resident in LDS; a K step = the three kx offsets of one (kz, ky) row
//                       (+ one idle slot); wave-level skip of the (kz, ky) rows no site of the 16-row block has.
//   spconv_cw_kernel    Cin >= 32: 128 rows x CT (<= 64) output columns per workgroup, planes streamed through double-buffered
//                       LDS in (kernel offset, 64-channel) units, codes gathered straight into MFMA operands (one bounded
//                       16-byte load per row and unit; no neighbour = out-of-range offset = zeros, no branch).
// MFMA orientation: A = weight plane (16 output channels x 64 K), B = codes (64 K x 16 rows) -> every lane ends up with four
// consecutive output channels of one row: 16-byte stores, 16-byte residual loads, 4-byte code stores.
#include "common.h"
#include <stdlib.h>

typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define CD_OOB 0xfffffff0u
#define CD_PLANES 3
#define CD_FRAC 65536.0f

// ------------------------------------------------------------------ quantiser + weight planes
struct SpPrepArgs {
    const float *x;           // [cap][C] producer output (fp32)
    const float *amax;        // [C] per-channel abs-max (dynamic: this frame's; static: calibrated)
    const int *count;
    int cap, C;
    float qmax;
    int per_tensor;
    int8_t *codes;            // [cap][C]
    const int8_t *w;          // [kvol][Cout][C] weight codes
    const float *w_dequant;   // [Cout]
    int8_t *wp;               // [kvol][Cout][3][C]  planes d0 d1 d2 of W''
    float *out_mul;           // [Cout]  S * t_o / 2^16
    int kvol, cout, row_blocks;
};

__global__ __launch_bounds__(256) void sp_prep_kernel(SpPrepArgs a) {
    __shared__ float s_scale[256];     // multiply: qmax / amax (0 for dead channels)
    __shared__ float s_dq[256];        // dequant step d_c = 1 / div
    __shared__ float s_S;
    const int C = a.C;
    if (threadIdx.x < C) {
        float am = a.amax[threadIdx.x];
        if (a.per_tensor) {
            am = 0.f;
            for (int c = 0; c < C; ++c) am = fmaxf(am, a.amax[c]);
        }
        const float eps = 1.0f / (float)(1 << 24);
        float scale = __fdiv_rn(a.qmax, am);
        float dv = scale;
        if (am <= eps) { scale = 0.f; dv = 1.f; }
        s_scale[threadIdx.x] = scale;
        s_dq[threadIdx.x] = scale == 0.f ? 0.f : __fdiv_rn(1.0f, dv);     // dead channel: every code is 0, its weights may be 0 too
    }
    __syncthreads();
    if ((int)blockIdx.x < a.row_blocks) {
        int n = *a.count;
        if (n > a.cap) n = a.cap;
        const int total4 = n * (C >> 2);
        for (int t = blockIdx.x * 256 + threadIdx.x; t < total4; t += a.row_blocks * 256) {
            const int c = (t * 4) & (C - 1);                 // C is a power of two (16 .. 256)
            const float4 xv = reinterpret_cast<const float4 *>(a.x)[t];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            signed char qc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float q = rintf(__fmul_rn(xs[k], s_scale[c + k]));
                q = fminf(fmaxf(q, -a.qmax - 1.f), a.qmax);
                qc[k] = (signed char)(int)q;
            }
            reinterpret_cast<char4 *>(a.codes)[t] = make_char4(qc[0], qc[1], qc[2], qc[3]);
        }
        return;
    }
    // weight blocks: S = largest live step; W'' = rint(code * (d_c / S) * 2^16) as three balanced base-256 digits
    if (threadIdx.x == 0) {
        float S = 0.f;
        for (int c = 0; c < C; ++c) S = fmaxf(S, s_dq[c]);
        s_S = S > 0.f ? S : 1.f;
    }
    __syncthreads();
    const float S = s_S;
    const int wb = (int)blockIdx.x - a.row_blocks, nwb = (int)gridDim.x - a.row_blocks;
    if (wb == 0)
        for (int o = threadIdx.x; o < a.cout; o += 256) a.out_mul[o] = __fmul_rn(__fmul_rn(S, a.w_dequant[o]), 1.0f / CD_FRAC);
    const int C8 = C >> 3, total8 = a.kvol * a.cout * C8;
    for (int t = wb * 256 + threadIdx.x; t < total8; t += nwb * 256) {
        const int ko = t / C8, c0 = (t - ko * C8) * 8;
        const int2 wv = *reinterpret_cast<const int2 *>(a.w + (size_t)ko * C + c0);
        unsigned d[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int code = (int)(signed char)((i < 4 ? wv.x : wv.y) >> (8 * (i & 3)));
            const float rel = __fdiv_rn(s_dq[c0 + i], S);
            int v = (int)rintf(__fmul_rn(__fmul_rn((float)code, rel), CD_FRAC));      // |v| <= 127 * 2^16 < 2^23: exact
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int dig = (int)(signed char)(v & 0xff);      // balanced digit in [-128, 127]
                d[p][i >> 2] |= (unsigned)(dig & 0xff) << (8 * (i & 3));
                v = (v - dig) >> 8;
            }
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint2 *>(a.wp + ((size_t)ko * 3 + p) * C + c0) = make_uint2(d[p][0], d[p][1]);
    }
}

// ------------------------------------------------------------------ conv kernels
struct SpCodedArgs {
    const int8_t *codes;      // [cap_in][CIN]
    const int8_t *wp;         // [kvol][COUT][3][CIN]
    const float *out_mul;     // [COUT]
    const float *bias, *post_scale, *post_shift, *residual;
    const int *nbr;           // [cap_out][kvol]
    const int *count_out;
    float *feats_out, *absmax_out;
    int cap_out, kvol, relu;
    unsigned codes_bytes;
    // static PTQ: the consumer's quantiser runs in this epilogue (its calibrated scale is a constant): codes of the output rows
    const float *next_scale;  // [COUT] qmax / amax_c of the consumer (0 for a dead channel), or nullptr
    int8_t *next_codes;       // [cap_out][COUT]
    float next_qmax;
    int skip_f32;             // nobody reads the fp32 rows (no residual / feature-map use): do not store them
};

// TensorQuantizer with a calibrated amax: clamp(rint(v * scale), -qmax - 1, qmax) — the arithmetic of sp_prep_kernel
__device__ __forceinline__ int cd_static_code(float v, float scale, float qmax) {
    float q = rintf(__fmul_rn(v, scale));
    q = fminf(fmaxf(q, -qmax - 1.f), qmax);
    return (int)q;
}

// XOR swizzle of the four 16-byte chunks of a 64-byte LDS row: conflict-free for the hardware's ds_read_b128 lane groups when
// lane l15 reads row l15 (csrc/conv2d.hip uses the same map)
__device__ __forceinline__ int cd_swz(int row, int chunk) { return chunk ^ ((0x1320 >> (((row >> 2) & 3) * 4)) & 3); }

// Per-channel epilogue operands staged in LDS: [5][CT] = out_mul, bias, post_scale, post_shift, next_scale
template <int CT>
__device__ __forceinline__ void cd_stage_epilogue(const SpCodedArgs &a, int ob, float *ep, int tid, int nth) {
    for (int t = tid; t < 5 * CT; t += nth) {
        const int f = t / CT, o = ob + (t - f * CT);
        float v = f == 0 ? a.out_mul[o] : (f == 1 ? (a.bias ? a.bias[o] : 0.f) : (f == 2 ? (a.post_scale ? a.post_scale[o] : 1.f)
                         : (f == 3 ? (a.post_shift ? a.post_shift[o] : 0.f) : (a.next_codes ? a.next_scale[o] : 0.f))));
        ep[t] = v;
    }
}

// One 16-channel x 16-row accumulator tile -> output: lane (l15 = row, g) holds channels 4g .. 4g+3. PL planes, plane 0 least
// significant of those multiplied (the multiplier carries 256^(3 - PL)). Returns the lane's |y| maxima of its four channels.
// Knock-out build of spconv_cw_kernel for scratch/cw_knockout.py (-DAQ3D_KO_CW=mask; timing only, results are wrong; the product
// build compiles none of it). bit 0: weight planes staged for the first unit only; 1: codes gathered for the first unit only;
// 2: one MFMA group of a unit's NT * PL * KS; 3: no fragment reads from LDS either; 4: no output stores; 5: no gather-table staging
// (every row sees its first unit's neighbours only). Every workgroup records where and when it ran.
#ifdef AQ3D_KO_CW
__device__ unsigned long long *g_cw_place;           // [gridDim.x][4]: XCC id, HW id, start, end (100 MHz wall clock); dead tiles stay 0
extern "C" int aq3d_diag_set_cw_place(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cw_place), &p, sizeof(p)); }
#define AQ3D_KO_CW_PLACE 1
#else
#define AQ3D_KO_CW 0
#endif

template <int PL, int CT, int COUT>
__device__ __forceinline__ f32x4v cd_emit_tile(const SpCodedArgs &a, const i32x4v (&acc)[PL], const float *ep, int cl, int ob, int m,
                                               bool live) {
    const f32x4v mul = *reinterpret_cast<const f32x4v *>(ep + 0 * CT + cl);
    const f32x4v bs = *reinterpret_cast<const f32x4v *>(ep + 1 * CT + cl);
    const f32x4v ps = *reinterpret_cast<const f32x4v *>(ep + 2 * CT + cl);
    const f32x4v pt = *reinterpret_cast<const f32x4v *>(ep + 3 * CT + cl);
    const f32x4v ns = *reinterpret_cast<const f32x4v *>(ep + 4 * CT + cl);
    f32x4v out = (f32x4v){0.f, 0.f, 0.f, 0.f};
    if (!live) return out;
    f32x4v res = (f32x4v){0.f, 0.f, 0.f, 0.f};
    if (a.residual) res = *reinterpret_cast<const f32x4v *>(a.residual + (size_t)m * COUT + ob + cl);
    int packed = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = (float)acc[PL - 1][r];
#pragma unroll
        for (int p = PL - 2; p >= 0; --p) s = __fmaf_rn(s, 256.0f, (float)acc[p][r]);      // exact up to fp32 rounding of the sum
        if (PL == 2) s = __fmul_rn(s, 256.0f);             // planes d2 d1 only: the multiplier is the three-plane one
        float v = __fmul_rn(s, mul[r]);
        if (a.bias) v = __fadd_rn(v, bs[r]);
        if (a.post_scale) v = __fadd_rn(__fmul_rn(v, ps[r]), pt[r]);
        if (a.residual) v = __fadd_rn(v, res[r]);
        if (a.relu) v = fmaxf(v, 0.f);
        out[r] = v;
        if (a.next_codes) packed |= (cd_static_code(v, ns[r], a.next_qmax) & 0xff) << (8 * r);
    }
    if (AQ3D_KO_CW & 16) asm volatile("" ::"v"(out));
    else if (!a.skip_f32) *reinterpret_cast<f32x4v *>(a.feats_out + (size_t)m * COUT + ob + cl) = out;
    if (a.next_codes) *reinterpret_cast<int *>(a.next_codes + (size_t)m * COUT + ob + cl) = packed;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = fabsf(out[r]);
    return out;
}

// max over the 16 lanes that share g (the rows of a tile), result in every lane
__device__ __forceinline__ float cd_max16(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64));
    v = fmaxf(v, __shfl_xor(v, 2, 64));
    v = fmaxf(v, __shfl_xor(v, 4, 64));
    v = fmaxf(v, __shfl_xor(v, 8, 64));
    return v;
}

// Cin = 16. K step q = kernel offsets 3q, 3q+1, 3q+2 (one (kz, ky) row of the 3x3x3 kernel) x 16 channels + an idle slot:
// lane (l15 = row, g) gathers the whole 16-byte code row of its neighbour at offset 3q + g.
template <int COUT, int NW, int PL>
__global__ __launch_bounds__(64 * NW) void spconv_cd_kernel(SpCodedArgs a) {
    constexpr int CIN = 16, KVOL = 27, NQ = 9;
    constexpr int NTH = 64 * NW, TMW = 16 * NW;
    constexpr int NT = COUT / 16;
    constexpr int LDR = NQ * 64 + 16;                     // bytes per LDS weight row (plane, channel): 9 K steps + pad
    constexpr int P0 = CD_PLANES - PL;                    // first plane multiplied
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *Wl = smem_raw;                                        // [PL][COUT][LDR]
    float *ep = reinterpret_cast<float *>(Wl + PL * COUT * LDR);         // [5][COUT]
    float *red = ep + 5 * COUT;                                          // [NW][COUT]
    constexpr int NPW = 16 * KVOL;                                       // gather-table entries of a wave's 16 rows: contiguous in memory
    int *nbr_s = reinterpret_cast<int *>(red + NW * COUT) + (threadIdx.x >> 6) * NPW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const int ntiles = (n + TMW - 1) / TMW;
    const auto rsrc_f = __builtin_amdgcn_make_buffer_rsrc((void *)a.codes, 0, (int)a.codes_bytes, 0x00020000);
    const auto rsrc_n = __builtin_amdgcn_make_buffer_rsrc((void *)a.nbr, 0, a.cap_out * KVOL * 4, 0x00020000);

    float amax_seen = (a.absmax_out && tid < COUT) ? a.absmax_out[tid] : 0.f;
    cd_stage_epilogue<COUT>(a, 0, ep, tid, NTH);
    // planes -> LDS: item = (plane, channel, K step, slot): 16 bytes = the 16 input channels of one kernel offset
    for (int t = tid; t < PL * COUT * NQ * 4; t += NTH) {
        const int s4 = t & 3, q = (t >> 2) % NQ, r = (t >> 2) / NQ;       // r = pl * COUT + o
        const int pl = r / COUT, o = r - pl * COUT;
        i32x4v v = (i32x4v){0, 0, 0, 0};
        if (s4 < 3) v = *reinterpret_cast<const i32x4v *>(a.wp + ((size_t)((3 * q + s4) * COUT + o) * CD_PLANES + P0 + pl) * CIN);
        *reinterpret_cast<i32x4v *>(Wl + r * LDR + q * 64 + s4 * 16) = v;
    }
    __syncthreads();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int rbase = tile * TMW + wave * 16;
        __syncthreads();                                 // the previous tile's reads of nbr_s are done
#pragma unroll
        for (int i = 0; i < (NPW + 63) / 64; ++i) {
            const int e = lane + 64 * i;
            if (NPW % 64 == 0 || e < NPW) {
                const bool live = rbase + e / KVOL < n;
                const int v = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc_n, live ? (unsigned)((rbase * KVOL + e) * 4) : CD_OOB, 0, 0);
                nbr_s[e] = live ? v : -1;
            }
        }
        __syncthreads();
        i32x4v acc[NT][PL];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int p = 0; p < PL; ++p) acc[i][p] = (i32x4v){0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = g < 3 ? nbr_s[l15 * KVOL + 3 * q + (g < 3 ? g : 0)] : -1;
            const unsigned off = idx >= 0 ? (unsigned)(idx * CIN) : CD_OOB;
            const i32x4v cv = __builtin_amdgcn_raw_buffer_load_b128(rsrc_f, off, 0, 0);   // no neighbour: out of range, no traffic
            // none of the wave's 16 rows has a neighbour in this (kz, ky) row: skip the LDS reads and the MFMAs
            if (__builtin_amdgcn_ballot_w64(idx >= 0) != 0ull) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int p = 0; p < PL; ++p) {
                        const i32x4v wf = *reinterpret_cast<const i32x4v *>(Wl + (p * COUT + nt * 16 + l15) * LDR + q * 64 + g * 16);
                        acc[nt][p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, cv, acc[nt][p], 0, 0, 0);
                    }
            }
        }
        // epilogue: lane (l15 = row, g) holds channels nt * 16 + 4g .. + 3
        const int m = rbase + l15;
        float cmax[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4v mx = cd_emit_tile<PL, COUT, COUT>(a, acc[nt], ep, nt * 16 + 4 * g, 0, m, m < n);
#pragma unroll
            for (int r = 0; r < 4; ++r) cmax[nt][r] = cd_max16(mx[r]);
        }
        if (a.absmax_out) {
            __syncthreads();
            if (l15 == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave * COUT + nt * 16 + 4 * g + r] = cmax[nt][r];
            }
            __syncthreads();
            if (tid < COUT) {       // COUT <= 64 * NW; guard read at kernel start (stale and lower only costs a redundant atomic)
                float mx = red[tid];
#pragma unroll
                for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w * COUT + tid]);
                if (mx > amax_seen) {
                    aq3d_atomic_max_pos(&a.absmax_out[tid], mx);
                    amax_seen = mx;
                }
            }
        }
    }
}

// In-kernel cycle stamps of spconv_cw_kernel for scratch/cw_stamps.py (a separate diagnostic build with -DAQ3D_STAMPS; the product
// library compiles none of it).
#ifdef AQ3D_STAMPS
__device__ unsigned long long *g_cw_stamps;          // [gridDim.x][4 waves][8]
extern "C" int aq3d_diag_set_stamps(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cw_stamps), &p, sizeof(p)); }
#define CW_NOW() ((unsigned long long)clock64())
#define CW_STAMP(var) const unsigned long long var = CW_NOW()
#define CW_ACC(acc, from, to) acc += (to) - (from)
#else
#define CW_STAMP(var)
#define CW_ACC(acc, from, to)
#endif

// Cin >= 32. Unit of work = 64 K elements: a 64-channel slice of one kernel offset (Cin >= 64) or TWO kernel offsets x 32
// channels (Cin = 32). The unit's planes go through LDS (double buffered, XOR-swizzled 64-byte rows), the codes of the wave's
// row tiles are gathered straight into MFMA B operands: ONE bounded 16-byte buffer load per row and unit.
// Workgroups per CU. The instantiations with two column blocks per row tile (Cout = 128: ~600 - 1000 live workgroups on the benchmark's
// levels) are compiled for THREE waves per SIMD (168 VGPRs, 8 spilled dwords outside the unit loop) and, at Cin = 128, stage half-offset
// units (64 K bytes per row: 41 instead of 65 KB of LDS), so that three workgroups share a CU: -3 % / -5 % of the family's time, -1 % frame
// latency, frame rate unchanged (profiles/r04/spconv_cw_ub64_ab.jsonl, spconv_cw_three_waves_ab.jsonl). The one-column-block shapes
// (<= 512 workgroups: everything is resident at two per CU anyway) lose 12 % to the tighter register budget and keep two.
#ifndef AQ3D_CW128_UB
#define AQ3D_CW128_UB 64        // A/B (scratch/ab_build.py): 128 = a unit is a whole kernel offset at Cin = 128 (half the barriers, two workgroups per CU)
#endif
template <int CIN, int COUT, int CT, int RT, int FD, int PL, int NW>
__global__ __launch_bounds__(64 * NW, (COUT / CT >= 2 && (CIN < 128 || AQ3D_CW128_UB == 64)) ? 3 : 2) void spconv_cw_kernel(SpCodedArgs a) {
    constexpr int NCB = COUT / CT;                      // column blocks: a workgroup computes CT of the COUT output channels
    constexpr int NTH = 64 * NW;                         // NW waves, RT 16-row tiles each
    constexpr int TM = NW * RT * 16;
    constexpr int NT = CT / 16;
    constexpr bool PAIR = CIN == 32;
    constexpr int UB = CIN >= 128 ? AQ3D_CW128_UB : 64;  // K bytes of a unit per row (64: a 64-channel slice of a kernel offset; 128 at Cin = 128: the
    constexpr int KS = UB / 64;                          // whole offset, full 128-byte code rows per gather and half the barriers — round 3's form)
    constexpr int NU = PAIR ? 1 : CIN / UB;
    constexpr int ROWS = PL * CT;                        // LDS rows of one unit: (plane, channel), UB bytes each
    constexpr int CPR = UB / 16;                         // 16-byte chunks per row
    constexpr int WCH = ROWS * CPR;                      // 16-byte chunks of one unit
    constexpr int NWL = (WCH + NTH - 1) / NTH;
    constexpr int P0 = CD_PLANES - PL;
    static_assert((CIN == 32 || CIN % 64 == 0) && CT % 16 == 0 && COUT % CT == 0, "channel granularity");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *Wb = smem_raw;                                          // [2 buffers][ROWS][64]
    int *nbr_s = reinterpret_cast<int *>(Wb + 2 * ROWS * UB);              // [TM][kvol]
    float *ep = reinterpret_cast<float *>(nbr_s + TM * 27);               // [5][CT]
    float *red = ep + 5 * CT;                                              // [NW][CT]
    int *mask_s = reinterpret_cast<int *>(red + NW * CT);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int kvol = a.kvol;
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    // column blocks of one row tile are dispatched to the SAME XCD (workgroups round-robin over the 8 XCDs): they gather the same rows
    const int bj = (int)blockIdx.x >> 3;
    const int cb = bj % NCB, ob = cb * CT;
    const int row0 = ((bj / NCB) * 8 + ((int)blockIdx.x & 7)) * TM;
    if (row0 >= n) return;
#ifdef AQ3D_KO_CW_PLACE
    const unsigned long long ko_t0 = wall_clock64();
#endif
    CW_STAMP(t_start);
#ifdef AQ3D_STAMPS
    unsigned long long s_cvt = 0, s_mfma = 0, s_store = 0, s_bar = 0;
#endif
    const auto rsrc_f = __builtin_amdgcn_make_buffer_rsrc((void *)a.codes, 0, (int)a.codes_bytes, 0x00020000);

    const float amax_seen = (a.absmax_out && tid < CT) ? a.absmax_out[ob + tid] : 0.f;      // guard of the final atomic max
    cd_stage_epilogue<CT>(a, ob, ep, tid, NTH);
    // gather-table tile -> LDS with 16-byte loads; the 27-bit "some row has this neighbour" mask is built from the loaded values
    // (OR over the wave by shuffles, one LDS word per wave) instead of a second pass over the staged table behind a barrier
    {
        const int rows = n - row0 < TM ? n - row0 : TM;
        const int lim = rows * kvol, total = TM * kvol;                       // total % 4 == 0 (TM a multiple of 128)
        const int alloc = (a.cap_out - row0 < TM ? a.cap_out - row0 : TM) * kvol;    // dwords of the table that exist from row0 on
        const int *src = a.nbr + (size_t)row0 * kvol;
        constexpr int NQ = (TM * 27 / 4 + NTH - 1) / NTH;
        i32x4v v[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int t4 = (tid + NTH * i) * 4;
            v[i] = (i32x4v){-1, -1, -1, -1};
            if (t4 + 3 < alloc) v[i] = *reinterpret_cast<const i32x4v *>(src + t4);
            else if (t4 < alloc) {
                v[i].x = src[t4];
                if (t4 + 1 < alloc) v[i].y = src[t4 + 1];
                if (t4 + 2 < alloc) v[i].z = src[t4 + 2];
            }
        }
        int bits = 0;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int t4 = (tid + NTH * i) * 4;
            if (t4 < total) {
                int kk = t4 % kvol;
                int e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (t4 + j >= lim) e[j] = -1;
                    if (e[j] >= 0) bits |= 1 << kk;
                    kk = kk + 1 == kvol ? 0 : kk + 1;
                }
                *reinterpret_cast<i32x4v *>(nbr_s + t4) = (i32x4v){e[0], e[1], e[2], e[3]};
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) bits |= __shfl_xor(bits, d, 64);
        if (lane == 0) mask_s[wave] = bits;
    }
    __syncthreads();
    int live = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) live |= mask_s[w];

    i32x4v acc[RT][NT][PL];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < PL; ++p) acc[rt][nt][p] = (i32x4v){0, 0, 0, 0};

    i32x4v A0[RT][KS], A1[RT][KS];
    // XOR swizzle of a row's 16-byte chunks, conflict-free for the ds_read_b128 lane groups when lane l15 reads row l15:
    // 64-byte rows: cd_swz; 128-byte rows: chunk ^ ((row >> 1) & 7)
    auto swz = [](int row, int chunk) { return UB == 64 ? cd_swz(row, chunk) : (chunk ^ ((row >> 1) & 7)); };
    // The planes of a unit go global -> LDS by LDS-DMA (no register round trip, no ds_write pass): the LDS image is lane-linear per
    // wave instruction (slot t = 16-byte chunk c of LDS row r = plane * CT + channel), so the XOR swizzle the fragment reads apply
    // sits on the SOURCE address: slot (r, c) receives chunk swz(r, c) of the row (cdna_hip_programming.md rule 21).
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    bool ko_first = true;                                // knock-out builds: the prologue's calls are the only ones that load
    auto stageW = [&](int b, int k, int h, int k2) {
        if (k >= kvol) return;                           // no next unit (wave-uniform): the idle unit multiplies zero codes
        if ((AQ3D_KO_CW & 1) && !ko_first) return;
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int t = tid + NTH * i;
            const int r = t / CPR, c = swz(r, t % CPR);
            const int pl = r / CT, o = r - pl * CT;
            const int kk = PAIR ? (c < 2 ? k : (k2 < kvol ? k2 : k)) : k;        // k2 = kvol: its codes are zeros, any bytes do
            const size_t off = PAIR ? (size_t)((kk * COUT + ob + o) * CD_PLANES + P0 + pl) * CIN + (c & 1) * 16
                                    : (size_t)((kk * COUT + ob + o) * CD_PLANES + P0 + pl) * CIN + h * UB + c * 16;
            if (WCH % NTH == 0 || t < WCH)
                __builtin_amdgcn_global_load_lds((gptr_t *)(a.wp + off), (lptr_t *)(Wb + (size_t)(b * ROWS) * UB + t * 16), 16, 0, 0);
        }
    };
    auto readSrc = [&](int (&src)[RT], int k) {
        const int kk = k < kvol ? k : 0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int v = nbr_s[(wave * (RT * 16) + rt * 16 + l15) * kvol + kk];
            src[rt] = k < kvol ? v : -1;
        }
    };
    // B operand of row tile rt, K step ks: lane (l15 = row, g) takes K bytes 64 ks + 16g .. + 15 of the unit
    auto loadA1 = [&](i32x4v (&A)[RT][KS], const int (&src)[RT], const int (&src2)[RT], int h, int q) {
        const int rt = q / KS, ks = q % KS;
        const int s = PAIR ? (g < 2 ? src[rt] : src2[rt]) : src[rt];
        const unsigned off = s >= 0 ? (unsigned)s * (unsigned)CIN + (unsigned)(PAIR ? (g & 1) * 16 : h * UB + ks * 64 + 16 * g) : CD_OOB;
        if ((AQ3D_KO_CW & 2) && !ko_first) return;
        A[rt][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_f, off, 0, 0);
    };
    auto loadA = [&](i32x4v (&A)[RT][KS], int k, int h, int k2) {
        int src[RT], src2[RT];
        readSrc(src, k);
        readSrc(src2, PAIR ? k2 : kvol);
#pragma unroll
        for (int q = 0; q < RT * KS; ++q) loadA1(A, src, src2, h, q);
    };
    auto next_live = [&](int k) {
        if (k >= kvol) return kvol;
        const int rest = live & ~((2 << k) - 1);
        return rest ? (__ffs(rest) - 1) : kvol;
    };
    auto next_unit = [&](int &k, int &h, int &k2) {
        if (k >= kvol) return;
        if (h + 1 < NU) { ++h; return; }
        h = 0;
        k = next_live(PAIR ? k2 : k);
        if (PAIR) k2 = next_live(k);
    };
    auto unit = [&](i32x4v (&Acur)[RT][KS], i32x4v (&Anext)[RT][KS], int b, int &k, int &h, int &k2) {
        CW_STAMP(ta);
        next_unit(k, h, k2);                             // (k, h, k2) now names the NEXT unit
        stageW(b ^ 1, k, h, k2);                         // its planes land in the other buffer while this unit multiplies
        CW_STAMP(tb);
        // fragment j = (nt * PL + plane) * KS + ks: read FD ahead of the MFMAs that use it, each feeds RT MFMAs
        constexpr int NF = NT * PL * KS;
        constexpr int NA = RT * KS;                          // gathers of the next unit
        constexpr int LA = NF - 1 < NA ? NF - 1 : NA;        // ... issued inside the MFMA loop (the rest after it)
        const unsigned char *bb = Wb + (size_t)(b * ROWS + l15) * UB;
        auto frag = [&](int j) {
            const int ks = j % KS, pl = (j / KS) % PL, nt = j / (KS * PL);
            if ((AQ3D_KO_CW & 8) && j > 0) return (i32x4v){j, l15, g, 1};
            return *reinterpret_cast<const i32x4v *>(bb + (pl * CT + nt * 16) * UB + swz(l15, ks * 4 + g) * 16);
        };
        i32x4v fb[FD + 1];
#pragma unroll
        for (int j = 0; j < FD && j < NF; ++j) fb[j] = frag(j);
        int src[RT], src2[RT];
        readSrc(src, k);
        readSrc(src2, PAIR ? k2 : kvol);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (j + FD < NF) fb[(j + FD) % (FD + 1)] = frag(j + FD);
            if (j >= 1 && j <= LA) loadA1(Anext, src, src2, h, j - 1);
            __builtin_amdgcn_sched_barrier(0);
            const int ks = j % KS, pl = (j / KS) % PL, nt = j / (KS * PL);
            const i32x4v f = fb[j % (FD + 1)];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if ((AQ3D_KO_CW & 4) && j > 0) asm volatile("" ::"v"(f), "v"(Acur[rt][ks]));
                else acc[rt][nt][pl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f, Acur[rt][ks], acc[rt][nt][pl], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = LA; q < NA; ++q) loadA1(Anext, src, src2, h, q);
        CW_STAMP(tc);
        CW_STAMP(td);
        __syncthreads();                                 // drains the LDS-DMA (vmcnt) and the next unit's gathers
        CW_STAMP(te);
        CW_ACC(s_cvt, ta, tb); CW_ACC(s_mfma, tb, tc); CW_ACC(s_store, tc, td); CW_ACC(s_bar, td, te);
    };

    CW_STAMP(t_pro);
    int k = live ? (__ffs(live) - 1) : kvol, h = 0;
    int k2 = PAIR ? next_live(k) : kvol;
    const int nunits = PAIR ? (__popc(live) + 1) / 2 : __popc(live) * NU;
    stageW(0, k, h, k2);
    loadA(A0, k, h, k2);
    ko_first = false;
    __syncthreads();
    CW_STAMP(t_fill);
    for (int u = 0; u < nunits; u += 2) {
        unit(A0, A1, 0, k, h, k2);
        unit(A1, A0, 1, k, h, k2);
    }
    CW_STAMP(t_loop);

    // ---- epilogue: lane (l15 = row of the tile, g) holds channels ob + nt * 16 + 4g .. + 3
    float cmax[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) cmax[nt][r] = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = row0 + wave * (RT * 16) + rt * 16 + l15;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4v mx = cd_emit_tile<PL, CT, COUT>(a, acc[rt][nt], ep, nt * 16 + 4 * g, ob, m, m < n);
#pragma unroll
            for (int r = 0; r < 4; ++r) cmax[nt][r] = fmaxf(cmax[nt][r], mx[r]);
        }
    }
    if (a.absmax_out) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) cmax[nt][r] = cd_max16(cmax[nt][r]);
        if (l15 == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * CT + nt * 16 + 4 * g + r] = cmax[nt][r];
        }
        __syncthreads();
        if (tid < CT) {     // CT <= 256; the guard value was read at kernel start (a stale, lower value only costs a redundant atomic)
            float mx = red[tid];
#pragma unroll
            for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w * CT + tid]);
            if (mx > amax_seen) aq3d_atomic_max_pos(&a.absmax_out[ob + tid], mx);
        }
    }
#ifdef AQ3D_KO_CW_PLACE
    if (tid == 0 && g_cw_place) {
        unsigned long long *o = g_cw_place + (size_t)blockIdx.x * 4;
        o[0] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        o[1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
        o[2] = ko_t0;
        o[3] = wall_clock64();
    }
#endif
#ifdef AQ3D_STAMPS
    if (lane == 0 && g_cw_stamps) {
        unsigned long long *o = g_cw_stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        const unsigned long long t_end = CW_NOW();
        o[0] = t_pro - t_start; o[1] = t_fill - t_pro; o[2] = s_cvt; o[3] = s_mfma; o[4] = s_store; o[5] = s_bar;
        o[6] = t_end - t_loop; o[7] = ((unsigned long long)nunits << 48) | (t_end - t_start);
    }
#endif
}

// ------------------------------------------------------------------ host side
template <int COUT, int NW, int PL>
static int launch_cd(const SpCodedArgs &a, hipStream_t stream) {
    constexpr size_t lds = (size_t)PL * COUT * (9 * 64 + 16) + (size_t)5 * COUT * 4 + (size_t)NW * COUT * 4 + (size_t)NW * 16 * 27 * 4;
    const int tiles = aq3d_div_up(a.cap_out, 16 * NW);
    const int grid = tiles < 2048 ? (tiles < 1 ? 1 : tiles) : 2048;
    auto kern = spconv_cd_kernel<COUT, NW, PL>;
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
    AQ3D_CHECK_LAUNCH("aq3d_spconv_coded(direct)");
    return AQ3D_OK;
}

template <int CIN, int COUT, int CT, int PL, int NW = 4>
static int launch_cw(const SpCodedArgs &a, hipStream_t stream) {
    constexpr int RT = 2, FD = 4;
    constexpr int TM = NW * 16 * RT;
    constexpr size_t lds = (size_t)2 * PL * CT * (CIN >= 128 ? AQ3D_CW128_UB : 64) + (size_t)TM * 27 * 4 + (size_t)(5 + NW) * CT * 4 + 64;
    const int grid = aq3d_div_up(aq3d_div_up(a.cap_out, TM), 8) * 8 * (COUT / CT);
    auto kern = spconv_cw_kernel<CIN, COUT, CT, RT, FD, PL, NW>;
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid < 1 ? 1 : grid), dim3(64 * NW), lds, stream, a);
    AQ3D_CHECK_LAUNCH("aq3d_spconv_coded(wide)");
    return AQ3D_OK;
}

extern "C" int aq3d_spconv_coded_supported(int32_t cin, int32_t cout, int32_t kvol) {
    if (kvol < 1 || kvol > 27) return 0;
    if (cin == 16) return kvol == 27 && (cout == 16 || cout == 32);
    return (cin == 32 && (cout == 32 || cout == 64)) || (cin == 64 && (cout == 64 || cout == 128)) || (cin == 128 && cout == 128);
}

extern "C" size_t aq3d_spconv_wprep_bytes(int32_t cin, int32_t cout, int32_t kvol) {
    return (size_t)kvol * cout * CD_PLANES * cin;
}

extern "C" int aq3d_spconv_prep(const float *x, const float *amax, const int32_t *count, int32_t cap, int32_t C, int32_t bits,
                                int32_t per_tensor, const int8_t *weight_codes, const float *w_dequant, int32_t kvol,
                                int32_t cout, int8_t *codes_out, void *wprep_out, float *out_mul_out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool rows = x != nullptr, weights = weight_codes != nullptr;
    AQ3D_REQUIRE(amax && (rows || weights), "aq3d_spconv_prep: null argument");
    AQ3D_REQUIRE(!rows || (count && codes_out), "aq3d_spconv_prep: the row quantiser needs count and codes_out");
    AQ3D_REQUIRE(!weights || (w_dequant && wprep_out && out_mul_out), "aq3d_spconv_prep: the weight planes need w_dequant, wprep_out and out_mul_out");
    AQ3D_REQUIRE(C >= 16 && C <= 256 && aq3d_is_pow2(C) && bits >= 2 && bits <= 8, "aq3d_spconv_prep: C=%d (a power of two in 16..256) bits=%d", C, bits);
    AQ3D_REQUIRE(cap > 0 && kvol >= 1 && kvol <= 27 && cout >= 1, "aq3d_spconv_prep: cap=%d kvol=%d cout=%d", cap, kvol, cout);
    AQ3D_REQUIRE((long long)cap * C < (1ll << 31), "aq3d_spconv_prep: cap * C overflows 32-bit indexing");
    SpPrepArgs a;
    a.x = x; a.amax = amax; a.count = count; a.cap = cap; a.C = C; a.qmax = (float)((1 << (bits - 1)) - 1);
    a.per_tensor = per_tensor; a.codes = codes_out; a.w = weight_codes; a.w_dequant = w_dequant; a.wp = (int8_t *)wprep_out;
    a.out_mul = out_mul_out; a.kvol = kvol; a.cout = cout;
    a.row_blocks = rows ? aq3d_grid((int64_t)cap * C / 4, 256) : 0;
    const int wblocks = weights ? aq3d_grid((int64_t)kvol * cout * C / 8, 256, 256) : 0;
    hipLaunchKernelGGL(sp_prep_kernel, dim3(a.row_blocks + wblocks), dim3(256), 0, stream, a);
    AQ3D_CHECK_LAUNCH("aq3d_spconv_prep");
    return AQ3D_OK;
}

extern "C" int aq3d_spconv_coded(const int8_t *codes, int32_t cap_in, const void *wprep, const float *out_mul, const float *bias,
                                 const float *post_scale, const float *post_shift, const float *residual, const int32_t *nbr,
                                 const int32_t *count_out, int32_t cap_out, int32_t cin, int32_t cout,
                                 int32_t kvol, int32_t flags, float *feats_out, float *absmax_out, const float *next_scale,
                                 int8_t *next_codes, int32_t next_bits, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(codes && wprep && out_mul && nbr && count_out && feats_out, "aq3d_spconv_coded: null argument");
    AQ3D_REQUIRE((post_scale == nullptr) == (post_shift == nullptr), "aq3d_spconv_coded: post_scale/post_shift must come together");
    AQ3D_REQUIRE(cap_in > 0 && cap_out > 0 && (long long)cap_in * cin < (1ll << 31), "aq3d_spconv_coded: cap_in=%d cap_out=%d", cap_in, cap_out);
    if (!aq3d_spconv_coded_supported(cin, cout, kvol)) {
        aq3d_set_error("aq3d_spconv_coded: unsupported shape cin=%d cout=%d kvol=%d", cin, cout, kvol);
        return AQ3D_ERR_UNSUPPORTED;
    }
    SpCodedArgs a;
    a.codes = codes; a.wp = (const int8_t *)wprep; a.out_mul = out_mul; a.bias = bias; a.post_scale = post_scale;
    a.post_shift = post_shift; a.residual = residual; a.nbr = nbr; a.count_out = count_out;
    a.feats_out = feats_out; a.absmax_out = absmax_out; a.cap_out = cap_out; a.kvol = kvol;
    a.relu = (flags & AQ3D_FLAG_RELU) ? 1 : 0;
    a.codes_bytes = (unsigned)((int64_t)cap_in * cin);
    AQ3D_REQUIRE((next_scale == nullptr) == (next_codes == nullptr), "aq3d_spconv_coded: next_scale / next_codes must come together");
    AQ3D_REQUIRE(!next_codes || (next_bits >= 2 && next_bits <= 8), "aq3d_spconv_coded: next_bits=%d", next_bits);
    AQ3D_REQUIRE(!(flags & AQ3D_FLAG_SKIP_F32) || next_codes, "aq3d_spconv_coded: AQ3D_FLAG_SKIP_F32 without next_codes would discard the output");
    a.next_scale = next_scale; a.next_codes = next_codes; a.next_qmax = next_codes ? (float)((1 << (next_bits - 1)) - 1) : 0.f;
    a.skip_f32 = (flags & AQ3D_FLAG_SKIP_F32) ? 1 : 0;
    // Cout = 128: two 64-column blocks per row tile (round 2, fp16 pipe, W150k, 35 k rows: 71 us against 98 us for one 128-column
    // workgroup and 66 us for four 32-column ones, which lose whole-frame throughput to the 4x repeated gathers)
#define AQ3D_CODED_DISPATCH(PL_) \
    do { \
        if (cin == 16 && cout == 16) return launch_cd<16, 4, PL_>(a, stream); \
        if (cin == 16 && cout == 32) return launch_cd<32, 8, PL_>(a, stream); \
        if (cin == 32 && cout == 32) return launch_cw<32, 32, 32, PL_>(a, stream); \
        if (cin == 32 && cout == 64) return launch_cw<32, 64, 64, PL_>(a, stream); \
        if (cin == 64 && cout == 64) return launch_cw<64, 64, 64, PL_>(a, stream); \
        if (cin == 64 && cout == 128) return launch_cw<64, 128, 64, PL_>(a, stream); \
        if (cin == 128 && cout == 128) return launch_cw<128, 128, 64, PL_>(a, stream); \
    } while (0)
    if (flags & AQ3D_FLAG_W16) AQ3D_CODED_DISPATCH(2);      // planes d2 d1 only (the kernels lift their sums by 256)
    AQ3D_CODED_DISPATCH(3);
#undef AQ3D_CODED_DISPATCH
    return AQ3D_ERR_UNSUPPORTED;
}
