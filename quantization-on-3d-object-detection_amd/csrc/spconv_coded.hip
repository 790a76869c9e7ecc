// Sparse conv on int8-CODED activations (1 byte per gathered element) with the per-input-channel activation scale folded into
// a per-frame fp16 weight copy — the production path of the QConvNd layers (quant/quant.py:36-58 around spconv).
//
// QConvNd quantises activations PER INPUT CHANNEL, so the scale d_c sits inside the K contraction:
//     y[m,o] = sum_{k,c} (q[nbr(m,k),c] * d_c) * (wq[k,o,c] * t_o).
// Here the gather moves the CODE q (int8, exact in fp16) and the scale moves into the weights:
//     W'[k,o,c] = wq[k,o,c] * (d_c / S),  S = max_c d_c,     y = (sum q * W') * (S * t_o)
// with W' held as an fp16 pair W' = hi + lo (hi = fp16(W'), lo = fp16(W' - hi): 22 significant bits; |W'| <= 127 so lo stays
// above the fp16 subnormal floor wherever it matters, and the matrix pipe honours subnormals). q * hi and q * lo are exact
// fp32 products of v_mfma_f32_16x16x32_f16 and ONE fp32 accumulator takes both. Against the round-1 format (activations as an
// fp16 hi|lo pair, 4 bytes per element and 4 gather instructions per row and 64-channel unit) the gather moves 4x fewer bytes
// with 4x fewer instructions at the same MFMA count; W' (27*C^2 elements) is rebuilt per frame by the quantiser launch.
//
//   sp_prep_kernel      one launch per layer: blocks [0, row_blocks) quantise the producer's fp32 rows to codes, the rest
//                       build W' (hi | lo planes) and the epilogue multipliers S * t_o. Both depend on the abs-max only.
//   spconv_cd_kernel    Cin = 16, kvol = 27: whole W' resident in LDS, per-lane 8-byte gathers straight into MFMA registers.
//   spconv_cw_kernel    Cin >= 32: 128 rows x CT (<= 64 at Cout = 128) output columns per workgroup, W' streamed through double-buffered LDS in
//                       (kernel offset, 64-channel) units, A fragments gathered into registers (one 16-byte load per row and unit).
// Code layout: rows of Cin bytes; for Cin >= 64 the bytes of every 64-channel group are permuted so that the 16 bytes a lane
// loads are the 8 + 8 channels of its two MFMA K chunks: position 16g + 8kc + i  <->  channel 32kc + 8g + i.
#include "common.h"
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x2v __attribute__((ext_vector_type(2)));
#define CD_OOB 0xfffffff0u

// 4 packed int8 -> 4 fp16 (exact) in 5 VALU ops: bias to unsigned, splice each byte under the fp16 exponent of 1024
// (0x64xx = 1024 + xx), subtract 1152 = 1024 + 128 with a packed add.
__device__ __forceinline__ void cd_cvt4(int w, f16x2 &lo, f16x2 &hi) {
    const unsigned t = (unsigned)w ^ 0x80808080u;
    const unsigned p0 = __builtin_amdgcn_perm(t, 0x64646464u, 0x00050004u);
    const unsigned p1 = __builtin_amdgcn_perm(t, 0x64646464u, 0x00070006u);
    const f16x2 bias = {(_Float16)1152.f, (_Float16)1152.f};
    lo = __builtin_bit_cast(f16x2, p0) - bias;
    hi = __builtin_bit_cast(f16x2, p1) - bias;
}
__device__ __forceinline__ f16x8 cd_cvt8(int w0, int w1) {
    f16x2 a, b, c, d;
    cd_cvt4(w0, a, b);
    cd_cvt4(w1, c, d);
    f16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1]; r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1];
    return r;
}

// position of channel c inside a code row (identity below 64 channels)
__host__ __device__ __forceinline__ int cd_code_pos(int c, int C) {
    if (C < 64) return c;
    const int grp = c >> 6, w = c & 63;
    return (grp << 6) + 16 * ((w & 31) >> 3) + 8 * (w >> 5) + (w & 7);
}

// ------------------------------------------------------------------ quantiser + W' builder
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
    _Float16 *wp;             // [kvol][Cout][2][C]   hi | lo planes
    float *out_mul;           // [Cout]  S * t_o
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
        const int C4 = C >> 2, total4 = n * C4;
        for (int t = blockIdx.x * 256 + threadIdx.x; t < total4; t += a.row_blocks * 256) {
            const int r = t / C4, c = (t - r * C4) * 4;
            const float4 xv = reinterpret_cast<const float4 *>(a.x)[t];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            signed char qc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float q = rintf(__fmul_rn(xs[k], s_scale[c + k]));
                q = fminf(fmaxf(q, -a.qmax - 1.f), a.qmax);
                qc[k] = (signed char)(int)q;
            }
            *reinterpret_cast<char4 *>(a.codes + (size_t)r * C + cd_code_pos(c, C)) = make_char4(qc[0], qc[1], qc[2], qc[3]);
        }
        return;
    }
    // weight blocks: S = largest live step; W' = code * (d_c / S) as hi | lo
    if (threadIdx.x == 0) {
        float S = 0.f;
        for (int c = 0; c < C; ++c) S = fmaxf(S, s_dq[c]);
        s_S = S > 0.f ? S : 1.f;
    }
    __syncthreads();
    const float S = s_S;
    const int wb = (int)blockIdx.x - a.row_blocks, nwb = (int)gridDim.x - a.row_blocks;
    if (wb == 0)
        for (int o = threadIdx.x; o < a.cout; o += 256) a.out_mul[o] = __fmul_rn(S, a.w_dequant[o]);
    const int C8 = C >> 3, total8 = a.kvol * a.cout * C8;
    for (int t = wb * 256 + threadIdx.x; t < total8; t += nwb * 256) {
        const int ko = t / C8, c0 = (t - ko * C8) * 8;
        const i32x2v wv = *reinterpret_cast<const i32x2v *>(a.w + (size_t)ko * C + c0);
        f16x8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int code = (int)(signed char)((i < 4 ? wv.x : wv.y) >> (8 * (i & 3)));
            const float rel = __fdiv_rn(s_dq[c0 + i], S);
            float v = __fmul_rn((float)code, rel);
            // v has to exist as ONE rounded fp32 value: without the barrier the backend folds (half)(code * rel) into a
            // mixed-precision fma (v_fma_mixlo_f16, a single rounding of the exact product) for one use of `h` and keeps
            // cvt(fp32 product) for the other — hi + lo then misses v by an fp16 ulp of hi (seen as 3e-2 absolute on |W'| ~ 100)
            asm volatile("" : "+v"(v));
            const _Float16 h = (_Float16)v;
            hi[i] = h;
            lo[i] = (_Float16)__fsub_rn(v, (float)h);
        }
        *reinterpret_cast<f16x8 *>(a.wp + ((size_t)ko * 2 + 0) * C + c0) = hi;
        *reinterpret_cast<f16x8 *>(a.wp + ((size_t)ko * 2 + 1) * C + c0) = lo;
    }
}

// ------------------------------------------------------------------ conv kernels
struct SpCodedArgs {
    const int8_t *codes;      // [cap_in][CIN]
    const _Float16 *wp;       // [kvol][COUT][2][CIN]
    const float *out_mul;     // [COUT]
    const float *bias, *post_scale, *post_shift, *residual;
    const int *nbr;           // [cap_out][kvol]
    const int *count_out;
    float *feats_out, *absmax_out;
    int cap_out, kvol, relu;
    unsigned codes_bytes;
    // static PTQ: the consumer's quantiser runs in this epilogue (its calibrated scale is a constant): codes of the output rows
    const float *next_scale;  // [COUT] qmax / amax_c of the consumer (0 for a dead channel), or nullptr
    int8_t *next_codes;       // [cap_out][COUT] in the consumer's code layout
    float next_qmax;
    int skip_f32;             // nobody reads the fp32 rows (no residual / feature-map use): do not store them
};

// TensorQuantizer with a calibrated amax: clamp(rint(v * scale), -qmax - 1, qmax) — the arithmetic of sp_prep_kernel
__device__ __forceinline__ signed char cd_static_code(float v, float scale, float qmax) {
    float q = rintf(__fmul_rn(v, scale));
    q = fminf(fmaxf(q, -qmax - 1.f), qmax);
    return (signed char)(int)q;
}

// Cin = 16: K chunk = 2 kernel offsets x 16 channels; lane (l15, g) holds 8 channels of offset 2q + (g >> 1).
template <int COUT, int KVOL, int NW, int PL>
__global__ __launch_bounds__(64 * NW) void spconv_cd_kernel(SpCodedArgs a) {
    constexpr int CIN = 16;
    constexpr int NTH = 64 * NW, TMW = 16 * NW;
    constexpr int NT = COUT / 16;
    constexpr int KT = KVOL * CIN;
    constexpr int NCH = (KT + 31) / 32;
    constexpr int LD = NCH * 32 + 8;                     // halves per LDS weight row (16-byte pad)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16 *Bh = reinterpret_cast<_Float16 *>(smem_raw);          // [COUT][LD] hi, K index = offset * CIN + channel
    _Float16 *Bl = Bh + COUT * LD;                                   // [COUT][LD] lo (PL == 2 only)
    float *red = reinterpret_cast<float *>(Bh + PL * COUT * LD);     // [NW][COUT]
    constexpr int NPW = 16 * KVOL;                                   // gather-table entries of a wave's 16 rows: contiguous in memory
    int *nbr_s = reinterpret_cast<int *>(red + NW * COUT) + (threadIdx.x >> 6) * NPW;     // [NW][16 * KVOL], this wave's slice

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const int ntiles = (n + TMW - 1) / TMW;
    const auto rsrc_f = __builtin_amdgcn_make_buffer_rsrc((void *)a.codes, 0, (int)a.codes_bytes, 0x00020000);
    const auto rsrc_n = __builtin_amdgcn_make_buffer_rsrc((void *)a.nbr, 0, a.cap_out * KVOL * 4, 0x00020000);

    // per-channel epilogue operands and the guard of the final atomic max: fetched before the weight staging, not at the tail
    float e_mul[NT], e_b[NT], e_ps[NT], e_pt[NT], e_nsc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int o = nt * 16 + l15;
        e_mul[nt] = a.out_mul[o];
        e_b[nt] = a.bias ? a.bias[o] : 0.f;
        e_ps[nt] = a.post_scale ? a.post_scale[o] : 1.f;
        e_pt[nt] = a.post_shift ? a.post_shift[o] : 0.f;
        e_nsc[nt] = a.next_codes ? a.next_scale[o] : 0.f;
    }
    float amax_seen = (a.absmax_out && tid < COUT) ? a.absmax_out[tid] : 0.f;
    for (int t = tid; t < PL * COUT * (LD / 8); t += NTH) {
        const int pl = t / (COUT * (LD / 8)), r = t % (COUT * (LD / 8));
        const int o = r / (LD / 8), kk = (r % (LD / 8)) * 8;
        i32x4v v = (i32x4v){0, 0, 0, 0};
        if (kk < KT) v = *reinterpret_cast<const i32x4v *>(a.wp + (((size_t)(kk / CIN) * COUT + o) * 2 + pl) * CIN + (kk % CIN));
        *reinterpret_cast<i32x4v *>((pl ? Bl : Bh) + o * LD + kk) = v;
    }
    __syncthreads();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // rulebook tile through LDS: the wave's 16 x KVOL gather-table entries are one contiguous block, loaded coalesced (7 dwords
        // per lane) instead of KVOL strided 4-byte loads per lane, of which a lane uses every other one
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
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int bofs = l15 * LD + 8 * g;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int j = 2 * q + (g >> 1);              // kernel offset of this lane's half of the K chunk
            const int idx = j < KVOL ? nbr_s[l15 * KVOL + (j < KVOL ? j : KVOL - 1)] : -1;
            const unsigned off = idx >= 0 ? (unsigned)(idx * CIN + (g & 1) * 8) : CD_OOB;
            const i32x2v cv = __builtin_amdgcn_raw_buffer_load_b64(rsrc_f, off, 0, 0);   // no neighbour: out of range, no traffic
            // none of the wave's 16 rows has a neighbour at these two offsets: skip the conversion, the LDS reads and the
            // MFMAs — adding exact zeros changes nothing
            if (__builtin_amdgcn_ballot_w64(idx >= 0) != 0ull) {
                const f16x8 fa = cd_cvt8(cv.x, cv.y);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f16x8 fh = *reinterpret_cast<const f16x8 *>(Bh + bofs + nt * 16 * LD + q * 32);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fh, acc[nt], 0, 0, 0);
                    if (PL == 2) {
                        const f16x8 fl = *reinterpret_cast<const f16x8 *>(Bl + bofs + nt * 16 * LD + q * 32);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fl, acc[nt], 0, 0, 0);
                    }
                }
            }
        }
        // epilogue (D layout: col = lane&15 -> out channel, row = 4g + reg)
        float cmax[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int o = nt * 16 + l15;
            const float mul = e_mul[nt], b = e_b[nt], ps = e_ps[nt], pt = e_pt[nt], nsc = e_nsc[nt];
            const int npos = cd_code_pos(o, COUT);
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tr = tile * TMW + wave * 16 + 4 * g + r;
                if (tr < n) {
                    const int m = tr;
                    float v = __fmul_rn(acc[nt][r], mul);
                    if (a.bias) v = __fadd_rn(v, b);
                    if (a.post_scale) v = __fadd_rn(__fmul_rn(v, ps), pt);
                    if (a.residual) v = __fadd_rn(v, a.residual[(size_t)m * COUT + o]);
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (!a.skip_f32) a.feats_out[(size_t)m * COUT + o] = v;
                    if (a.next_codes) a.next_codes[(size_t)m * COUT + npos] = cd_static_code(v, nsc, a.next_qmax);
                    mx = fmaxf(mx, fabsf(v));
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            cmax[nt] = mx;
        }
        if (a.absmax_out) {
            __syncthreads();
            if (g == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) red[wave * COUT + nt * 16 + l15] = cmax[nt];
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

// Cin >= 32. Unit of work = (kernel offset k, 64-channel slice h) [32 channels when Cin = 32]: the two W' planes of the
// slice go through LDS (double buffered, XOR-swizzled 128-byte rows as in the round-1 wide kernel), the A fragments of the
// wave's two 16-row tiles are gathered straight into registers: ONE bounded buffer load per row and unit (16 bytes = the
// 8 + 8 channels of the lane's two K chunks; 8 bytes when Cin = 32), missing neighbour = out-of-range offset = zeros.
template <int CIN, int COUT, int CT, int RT, int FD, int PL>
__global__ __launch_bounds__(256) void spconv_cw_kernel(SpCodedArgs a) {
    constexpr int NCB = COUT / CT;                      // column blocks: a workgroup computes CT of the COUT output channels
    constexpr int TM = 4 * RT * 16;
    constexpr int NT = CT / 16;
    // A unit is 64 K elements: a 64-channel slice of one kernel offset (Cin >= 64) or TWO kernel offsets x 32 channels (Cin = 32:
    // K chunk 0 = first offset, chunk 1 = second; half as many staging rounds and barriers as one offset per unit).
    constexpr bool PAIR = CIN == 32;
    constexpr int UC = 64;
    constexpr int NU = PAIR ? 1 : CIN / UC;
    constexpr int KC = 2;
    constexpr bool SWZ = true;
    constexpr int LDW = UC;                              // halves per LDS weight row of one plane (XOR-swizzled 16-byte chunks)
    constexpr int PCH = CT * UC / 8;                  // 16-byte chunks of one plane of a unit
    constexpr int WCH = PL * PCH;                        // PL = 2: hi | lo planes of W'; 1: the hi plane only (AQ3D_FLAG_W_FP16)
    constexpr int NWL = (WCH + 255) / 256;
    static_assert((CIN == 32 || CIN % 64 == 0) && CT % 16 == 0 && COUT % CT == 0, "channel granularity");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16 *Wb = reinterpret_cast<_Float16 *>(smem_raw);              // [2 buffers][PL planes][CT][LDW]
    int *nbr_s = reinterpret_cast<int *>(Wb + 2 * PL * CT * LDW);          // [TM][kvol]
    float *red = reinterpret_cast<float *>(nbr_s + TM * 27);            // [4][CT]
    int *mask_s = reinterpret_cast<int *>(red + 4 * CT);

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
    CW_STAMP(t_start);
#ifdef AQ3D_STAMPS
    unsigned long long s_cvt = 0, s_mfma = 0, s_store = 0, s_bar = 0;
#endif
    const auto rsrc_f = __builtin_amdgcn_make_buffer_rsrc((void *)a.codes, 0, (int)a.codes_bytes, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.wp, 0, kvol * COUT * 2 * CIN * 2, 0x00020000);

    // the per-channel epilogue operands are fetched now: loading them after the unit loop put one more global round trip at the
    // end of every workgroup (stamps, scratch/cw_stamps.py: prologue + epilogue were 45 % of the 32-channel kernel, 34 % at 64)
    const float amax_seen = (a.absmax_out && tid < CT) ? a.absmax_out[ob + tid] : 0.f;      // guard of the final atomic max
    float e_mul[NT], e_bs[NT], e_ps[NT], e_pt[NT], e_nsc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int o = ob + nt * 16 + l15;
        e_mul[nt] = a.out_mul[o];
        e_bs[nt] = a.bias ? a.bias[o] : 0.f;
        e_ps[nt] = a.post_scale ? a.post_scale[o] : 1.f;
        e_pt[nt] = a.post_shift ? a.post_shift[o] : 0.f;
        e_nsc[nt] = a.next_codes ? a.next_scale[o] : 0.f;
    }
    // gather-table tile -> LDS with 16-byte loads; the 27-bit "some row has this neighbour" mask is built from the loaded values
    // (OR over the wave by shuffles, one LDS word per wave) instead of a second pass over the staged table behind a barrier
    {
        const int rows = n - row0 < TM ? n - row0 : TM;
        const int lim = rows * kvol, total = TM * kvol;                       // total % 4 == 0 (TM = 128)
        const int alloc = (a.cap_out - row0 < TM ? a.cap_out - row0 : TM) * kvol;    // dwords of the table that exist from row0 on
        const int *src = a.nbr + (size_t)row0 * kvol;
        constexpr int NQ = (TM * 27 / 4 + 255) / 256;
        i32x4v v[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int t4 = (tid + 256 * i) * 4;
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
            const int t4 = (tid + 256 * i) * 4;
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
    const int live = mask_s[0] | mask_s[1] | mask_s[2] | mask_s[3];

    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    i32x4v A0[RT], A1[RT], wn[NWL];
    // chunk t of a unit: plane = t / PCH, row o, 16-byte chunk c8 of the UC-channel slice
    auto loadW = [&](int k, int h, int k2) {
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int t = tid + 256 * i;
            const int pl = t / PCH, r = t % PCH;
            const int c8 = r % (UC / 8), o = r / (UC / 8);
            const unsigned off = PAIR ? (unsigned)(((((c8 < 4 ? k : k2) * COUT + ob + o) * 2 + pl) * CIN + (c8 & 3) * 8) * 2)
                                      : (unsigned)((((k * COUT + ob + o) * 2 + pl) * CIN + h * UC + c8 * 8) * 2);
            if (WCH % 256 == 0 || t < WCH)
                wn[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, off, 0, 0);   // k = kvol is past the end of the tensor: zeros
        }
    };
    auto storeW = [&](int b) {
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int t = tid + 256 * i;
            const int pl = t / PCH, r = t % PCH;
            const int c8 = r % (UC / 8), o = r / (UC / 8);
            if (WCH % 256 == 0 || t < WCH)
                *reinterpret_cast<i32x4v *>(Wb + ((b * PL + pl) * CT + o) * LDW + (SWZ ? (c8 ^ ((o >> 1) & 7)) : c8) * 8) = wn[i];
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
    auto loadA1 = [&](i32x4v (&A)[RT], const int (&src)[RT], const int (&src2)[RT], int h, int rt) {
        if (!PAIR) {
            const unsigned off = src[rt] >= 0 ? (unsigned)src[rt] * (unsigned)CIN + (unsigned)(h * 64 + 16 * g) : CD_OOB;
            A[rt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_f, off, 0, 0);
        } else {    // 8 channels of the lane's K group from each of the unit's two kernel offsets
            const unsigned off = src[rt] >= 0 ? (unsigned)src[rt] * (unsigned)CIN + (unsigned)(8 * g) : CD_OOB;
            const unsigned off2 = src2[rt] >= 0 ? (unsigned)src2[rt] * (unsigned)CIN + (unsigned)(8 * g) : CD_OOB;
            const i32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrc_f, off, 0, 0);
            const i32x2v v2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc_f, off2, 0, 0);
            A[rt] = (i32x4v){v.x, v.y, v2.x, v2.y};
        }
    };
    auto loadA = [&](i32x4v (&A)[RT], int k, int h, int k2) {
        int src[RT], src2[RT];
        readSrc(src, k);
        readSrc(src2, PAIR ? k2 : kvol);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) loadA1(A, src, src2, h, rt);
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
    auto unit = [&](i32x4v (&Acur)[RT], i32x4v (&Anext)[RT], int b, int &k, int &h, int &k2) {
        CW_STAMP(ta);
        next_unit(k, h, k2);                             // (k, h, k2) now names the NEXT unit
        loadW(k, h, k2);
        f16x8 F[RT][KC];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            F[rt][0] = cd_cvt8(Acur[rt].x, Acur[rt].y);
            F[rt][1] = cd_cvt8(Acur[rt].z, Acur[rt].w);
        }
#ifdef AQ3D_STAMPS
        asm volatile("" :: "v"(F[0][0]), "v"(F[RT - 1][1]));
#endif
        CW_STAMP(tb);
        // fragment j = (kc * NT + nt) * PL + plane: read FD ahead of the MFMAs that use it, each feeds RT MFMAs
        constexpr int NF = KC * NT * PL;
        constexpr int LA = NF - 1 < RT ? NF - 1 : RT;        // next unit's A loads issued inside the MFMA loop (the rest after it)
        const int c0 = SWZ ? (g ^ ((l15 >> 1) & 7)) : g;
        const _Float16 *bb = Wb + (b * PL * CT + l15) * LDW;
        auto frag = [&](int j) {
            const int pl = j % PL, nt = (j / PL) % NT, kc = (j / PL) / NT;
            const int c = SWZ ? (c0 ^ (kc * 4)) : (c0 + kc * 4);
            return *reinterpret_cast<const f16x8 *>(bb + (pl * CT + nt * 16) * LDW + c * 8);
        };
        f16x8 fb[FD + 1];
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
            const int nt = (j / PL) % NT, kc = (j / PL) / NT;
            const f16x8 f = fb[j % (FD + 1)];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F[rt][kc], f, acc[rt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int rt = LA; rt < RT; ++rt) loadA1(Anext, src, src2, h, rt);
        CW_STAMP(tc);
        storeW(b ^ 1);
        CW_STAMP(td);
        __syncthreads();
        CW_STAMP(te);
        CW_ACC(s_cvt, ta, tb); CW_ACC(s_mfma, tb, tc); CW_ACC(s_store, tc, td); CW_ACC(s_bar, td, te);
    };

    CW_STAMP(t_pro);
    int k = live ? (__ffs(live) - 1) : kvol, h = 0;
    int k2 = PAIR ? next_live(k) : kvol;
    const int nunits = PAIR ? (__popc(live) + 1) / 2 : __popc(live) * NU;
    loadW(k, h, k2);
    loadA(A0, k, h, k2);
    storeW(0);
    __syncthreads();
    CW_STAMP(t_fill);
    for (int u = 0; u < nunits; u += 2) {
        unit(A0, A1, 0, k, h, k2);
        unit(A1, A0, 1, k, h, k2);
    }
    CW_STAMP(t_loop);

    // ---- epilogue (D layout: col = lane&15 -> out channel, row = 4g + reg -> tile row)
    int mrow[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tr = row0 + wave * (RT * 16) + rt * 16 + 4 * g + r;
            mrow[rt][r] = tr < n ? tr : -1;
        }
    float cmax[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int o = ob + nt * 16 + l15;
        const float mul = e_mul[nt], bs = e_bs[nt], ps = e_ps[nt], pt = e_pt[nt], nsc = e_nsc[nt];
        const int npos = cd_code_pos(o, COUT);
        float mx = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow[rt][r];
                if (m >= 0) {
                    float v = __fmul_rn(acc[rt][nt][r], mul);
                    if (a.bias) v = __fadd_rn(v, bs);
                    if (a.post_scale) v = __fadd_rn(__fmul_rn(v, ps), pt);
                    if (a.residual) v = __fadd_rn(v, a.residual[(size_t)m * COUT + o]);
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (!a.skip_f32) a.feats_out[(size_t)m * COUT + o] = v;
                    if (a.next_codes) a.next_codes[(size_t)m * COUT + npos] = cd_static_code(v, nsc, a.next_qmax);
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
            for (int nt = 0; nt < NT; ++nt) red[wave * CT + nt * 16 + l15] = cmax[nt];
        }
        __syncthreads();
        if (tid < CT) {     // CT <= 256; the guard value was read at kernel start (a stale, lower value only costs a redundant atomic)
            const float mx = fmaxf(fmaxf(red[tid], red[CT + tid]), fmaxf(red[2 * CT + tid], red[3 * CT + tid]));
            if (mx > amax_seen) aq3d_atomic_max_pos(&a.absmax_out[ob + tid], mx);
        }
    }
#ifdef AQ3D_STAMPS
    if (lane == 0 && g_cw_stamps) {
        unsigned long long *o = g_cw_stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
        const unsigned long long t_end = CW_NOW();
        o[0] = t_pro - t_start; o[1] = t_fill - t_pro; o[2] = s_cvt; o[3] = s_mfma; o[4] = s_store; o[5] = s_bar;
        o[6] = t_end - t_loop; o[7] = ((unsigned long long)nunits << 48) | (t_end - t_start);
    }
#endif
}

// ------------------------------------------------------------------ host side
template <int COUT, int KVOL, int NW, int PL>
static int launch_cd(const SpCodedArgs &a, hipStream_t stream) {
    constexpr int NCH = (KVOL * 16 + 31) / 32;
    constexpr size_t lds = (size_t)PL * COUT * (NCH * 32 + 8) * 2 + NW * COUT * 4 + (size_t)NW * 16 * KVOL * 4;
    const int tiles = aq3d_div_up(a.cap_out, 16 * NW);
    const int grid = tiles < 2048 ? (tiles < 1 ? 1 : tiles) : 2048;
    auto kern = spconv_cd_kernel<COUT, KVOL, NW, PL>;
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
    AQ3D_CHECK_LAUNCH("aq3d_spconv_coded(direct)");
    return AQ3D_OK;
}

template <int CIN, int COUT, int CT, int PL>
static int launch_cw(const SpCodedArgs &a, hipStream_t stream) {
    // 2 row tiles per wave (RT), fragments read 4 ahead (FD): measured round 2 against RT = 1 / 4 and FD = 8 (all slower; RT = 1 also
    // at Cin = 32: 29.4 against 27.9 us). W' loads two units ahead (second register set): no change (71.9 / 47.0 / 28.4 us) — removed.
    constexpr int RT = 2, FD = 4;
    constexpr int TM = 64 * RT;
    constexpr size_t lds = (size_t)2 * PL * CT * 64 * 2 + (size_t)TM * 27 * 4 + (size_t)4 * CT * 4 + 64;
    const int grid = aq3d_div_up(aq3d_div_up(a.cap_out, TM), 8) * 8 * (COUT / CT);
    auto kern = spconv_cw_kernel<CIN, COUT, CT, RT, FD, PL>;
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid < 1 ? 1 : grid), dim3(256), lds, stream, a);
    AQ3D_CHECK_LAUNCH("aq3d_spconv_coded(wide)");
    return AQ3D_OK;
}

extern "C" int aq3d_spconv_coded_supported(int32_t cin, int32_t cout, int32_t kvol) {
    if (kvol < 1 || kvol > 27) return 0;
    if (cin == 16) return kvol == 27 && (cout == 16 || cout == 32);
    return (cin == 32 && (cout == 32 || cout == 64)) || (cin == 64 && (cout == 64 || cout == 128)) || (cin == 128 && cout == 128);
}

extern "C" size_t aq3d_spconv_wprep_bytes(int32_t cin, int32_t cout, int32_t kvol) {
    return (size_t)kvol * cout * 2 * cin * sizeof(_Float16);
}

extern "C" int aq3d_spconv_prep(const float *x, const float *amax, const int32_t *count, int32_t cap, int32_t C, int32_t bits,
                                int32_t per_tensor, const int8_t *weight_codes, const float *w_dequant, int32_t kvol,
                                int32_t cout, int8_t *codes_out, void *wprep_out, float *out_mul_out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool rows = x != nullptr, weights = weight_codes != nullptr;
    AQ3D_REQUIRE(amax && (rows || weights), "aq3d_spconv_prep: null argument");
    AQ3D_REQUIRE(!rows || (count && codes_out), "aq3d_spconv_prep: the row quantiser needs count and codes_out");
    AQ3D_REQUIRE(!weights || (w_dequant && wprep_out && out_mul_out), "aq3d_spconv_prep: the weight copy needs w_dequant, wprep_out and out_mul_out");
    AQ3D_REQUIRE(C >= 16 && C <= 256 && (C & 15) == 0 && bits >= 2 && bits <= 8, "aq3d_spconv_prep: C=%d bits=%d", C, bits);
    AQ3D_REQUIRE(cap > 0 && kvol >= 1 && kvol <= 27 && cout >= 1, "aq3d_spconv_prep: cap=%d kvol=%d cout=%d", cap, kvol, cout);
    AQ3D_REQUIRE((long long)cap * C < (1ll << 31), "aq3d_spconv_prep: cap * C overflows 32-bit indexing");
    SpPrepArgs a;
    a.x = x; a.amax = amax; a.count = count; a.cap = cap; a.C = C; a.qmax = (float)((1 << (bits - 1)) - 1);
    a.per_tensor = per_tensor; a.codes = codes_out; a.w = weight_codes; a.w_dequant = w_dequant; a.wp = (_Float16 *)wprep_out;
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
    a.codes = codes; a.wp = (const _Float16 *)wprep; a.out_mul = out_mul; a.bias = bias; a.post_scale = post_scale;
    a.post_shift = post_shift; a.residual = residual; a.nbr = nbr; a.count_out = count_out;
    a.feats_out = feats_out; a.absmax_out = absmax_out; a.cap_out = cap_out; a.kvol = kvol;
    a.relu = (flags & AQ3D_FLAG_RELU) ? 1 : 0;
    a.codes_bytes = (unsigned)((int64_t)cap_in * cin);
    AQ3D_REQUIRE((next_scale == nullptr) == (next_codes == nullptr), "aq3d_spconv_coded: next_scale / next_codes must come together");
    AQ3D_REQUIRE(!next_codes || (next_bits >= 2 && next_bits <= 8), "aq3d_spconv_coded: next_bits=%d", next_bits);
    AQ3D_REQUIRE(!(flags & AQ3D_FLAG_SKIP_F32) || next_codes, "aq3d_spconv_coded: AQ3D_FLAG_SKIP_F32 without next_codes would discard the output");
    a.next_scale = next_scale; a.next_codes = next_codes; a.next_qmax = next_codes ? (float)((1 << (next_bits - 1)) - 1) : 0.f;
    a.skip_f32 = (flags & AQ3D_FLAG_SKIP_F32) ? 1 : 0;
    // Cout = 128: two 64-column blocks per row tile (measured on W150k, 35 k rows: 71 us against 98 us for one 128-column workgroup
    // and 66 us for four 32-column ones, which lose whole-frame throughput to the 4x repeated gathers: 599 against 614 frames/s)
#define AQ3D_CODED_DISPATCH(PL_) \
    do { \
        if (cin == 16 && cout == 16) return launch_cd<16, 27, 4, PL_>(a, stream); \
        if (cin == 16 && cout == 32) return launch_cd<32, 27, 8, PL_>(a, stream);    /* 58 KB of W' per workgroup: 8 waves share it */ \
        /* (32 -> 32 with the whole W' resident like Cin = 16 — 110 KB, one workgroup per CU — measured 41 us against 30 us) */ \
        if (cin == 32 && cout == 32) return launch_cw<32, 32, 32, PL_>(a, stream); \
        if (cin == 32 && cout == 64) return launch_cw<32, 64, 64, PL_>(a, stream); \
        if (cin == 64 && cout == 64) return launch_cw<64, 64, 64, PL_>(a, stream); \
        if (cin == 64 && cout == 128) return launch_cw<64, 128, 64, PL_>(a, stream); \
        if (cin == 128 && cout == 128) return launch_cw<128, 128, 64, PL_>(a, stream); \
    } while (0)
    if (flags & AQ3D_FLAG_W_FP16) AQ3D_CODED_DISPATCH(1);
    AQ3D_CODED_DISPATCH(2);
#undef AQ3D_CODED_DISPATCH
    return AQ3D_ERR_UNSUPPORTED;
}
