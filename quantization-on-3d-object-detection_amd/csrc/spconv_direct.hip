// Direct-gather sparse conv for narrow layers (Cin 16 / 32): no LDS staging of activations, no barriers in the K loop.
//
// At C = 16/32 the whole weight tensor of the layer is small (27*C*C fp16 = 14-56 KB), so each workgroup stages it into LDS
// ONCE and keeps it across all its tiles (persistent grid). The MFMA A operand of v_mfma_f32_16x16x32_f16 is, per lane,
// 8 consecutive K values of one output row — i.e. 16 contiguous bytes of ONE gathered feature row — so every lane fetches
// its fragment straight from global/L2 with one hardware-bounded 16-byte buffer load per (hi, lo) half: the K index
// (kernel offset, channel) is laid out so that a 32-wide K chunk covers 2 offsets x 16 channels (Cin 16) or 1 offset x 32
// channels (Cin 32). Missing neighbours are out-of-range offsets → zeros, no branch. Waves never synchronise inside a tile,
// the fully unrolled chunk loop keeps several gathers in flight per lane, and occupancy (not a barrier) hides L2 latency.
// Numerics are those of spconv_split.hip: exact fp16 products, fp32 accumulation, acc = hi-sum + lo-sum / 2048.
#include "common.h"
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
#define DG_OOB 0xfffffff0u

struct SpconvDirectArgs {
    const _Float16 *feats;    // [M_in, 2*CIN] hi | lo
    const _Float16 *weight;   // [kvol][COUT][CIN] integer codes as fp16
    const float *w_dequant, *bias, *post_scale, *post_shift, *residual;
    const int *nbr;           // [cap_out][kvol]
    const int *count_out;
    float *feats_out, *absmax_out;
    int cap_out, kvol, relu;
    unsigned feats_bytes;
};

// NW waves per workgroup share the LDS-resident weights; the kernel is bound by gather latency (rocprof: 83 % of the wave
// cycles parked in s_waitcnt), so what matters is how many independent waves a CU holds: the weights cap it at 2 workgroups.
template <int CIN, int COUT, int KVOL, int NW>
__global__ __launch_bounds__(64 * NW) void spconv_direct_kernel(SpconvDirectArgs a) {
    constexpr int NTH = 64 * NW, TMW = 16 * NW;
    constexpr int NT = COUT / 16;
    constexpr int KT = KVOL * CIN;                       // total K
    constexpr int NCH = (KT + 31) / 32;                  // 32-wide K chunks
    constexpr int LD = NCH * 32 + 8;                     // halves per LDS weight row
    constexpr int OPC = 32 / CIN;                        // kernel offsets per chunk (2 for Cin 16, 1 for Cin 32)
    static_assert(CIN == 16 || CIN == 32, "narrow layers only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16 *Bs = reinterpret_cast<_Float16 *>(smem_raw);          // [COUT][LD], K index = offset * CIN + channel
    float *red = reinterpret_cast<float *>(Bs + COUT * LD);          // [4][COUT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    int n = *a.count_out;
    if (n > a.cap_out) n = a.cap_out;
    const int ntiles = (n + TMW - 1) / TMW;
    const auto rsrc_f = __builtin_amdgcn_make_buffer_rsrc((void *)a.feats, 0, (int)a.feats_bytes, 0x00020000);
    const auto rsrc_n = __builtin_amdgcn_make_buffer_rsrc((void *)a.nbr, 0, a.cap_out * KVOL * 4, 0x00020000);

    // stage the whole weight tensor once: global [k][o][c] → Bs[o][k*CIN + c]; zero the K padding
    for (int t = tid; t < COUT * (LD / 8); t += NTH) {
        const int o = t / (LD / 8), c8 = t % (LD / 8);
        const int kk = c8 * 8;                                       // K index of this 16-byte chunk
        i32x4v v = (i32x4v){0, 0, 0, 0};
        if (kk < KT) v = *reinterpret_cast<const i32x4v *>(a.weight + ((size_t)(kk / CIN) * COUT + o) * CIN + (kk % CIN));
        *reinterpret_cast<i32x4v *>(Bs + o * LD + kk) = v;
    }
    __syncthreads();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row = tile * TMW + wave * 16 + l15;                 // this lane's output row (4 g-groups share it)
        // the row's KVOL neighbour indices, one dword load each (out-of-range rows read as 0 → treated as "no neighbour")
        int nb[KVOL];
#pragma unroll
        for (int k = 0; k < KVOL; ++k) {
            const unsigned off = row < n ? (unsigned)((row * KVOL + k) * 4) : DG_OOB;
            nb[k] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc_n, off, 0, 0);
        }
        if (row >= n) {
#pragma unroll
            for (int k = 0; k < KVOL; ++k) nb[k] = -1;
        }
        f32x4 acc_hi[NT], acc_lo[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            acc_hi[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc_lo[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const _Float16 *bb = Bs + l15 * LD + 8 * g;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            // lane's 8 K values of chunk q: offset j, channels cb..cb+7
            int idx;
            int cb;
            if (OPC == 2) {
                const int j0 = 2 * q, j1 = 2 * q + 1;
                const int i0 = nb[j0 < KVOL ? j0 : KVOL - 1];
                const int i1 = j1 < KVOL ? nb[j1 < KVOL ? j1 : KVOL - 1] : -1;
                idx = (g >> 1) ? i1 : i0;
                cb = (g & 1) * 8;
            } else {
                idx = nb[q < KVOL ? q : KVOL - 1];
                cb = g * 8;
            }
            const unsigned off = idx >= 0 ? (unsigned)((idx * 2 * CIN + cb) * 2) : DG_OOB;
            const i32x4v vh = __builtin_amdgcn_raw_buffer_load_b128(rsrc_f, off, 0, 0);
            const i32x4v vl = __builtin_amdgcn_raw_buffer_load_b128(rsrc_f, off, CIN * 2, 0);   // lo half: +CIN halves (soffset)
            const f16x8 fh = __builtin_bit_cast(f16x8, vh), fl = __builtin_bit_cast(f16x8, vl);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f16x8 fb = *reinterpret_cast<const f16x8 *>(bb + nt * 16 * LD + q * 32);
                acc_hi[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, fb, acc_hi[nt], 0, 0, 0);
                acc_lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, fb, acc_lo[nt], 0, 0, 0);
            }
        }

        // epilogue (D layout: col = lane&15 → out channel, row = 4g + reg → row of this wave's 16)
        float cmax[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int o = nt * 16 + l15;
            const float wdq = a.w_dequant[o];
            const float b = a.bias ? a.bias[o] : 0.f;
            const float ps = a.post_scale ? a.post_scale[o] : 1.f;
            const float pt = a.post_shift ? a.post_shift[o] : 0.f;
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = tile * TMW + wave * 16 + 4 * g + r;
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
            __syncthreads();                                         // red from the previous tile has been consumed
            if (g == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) red[wave * COUT + nt * 16 + l15] = cmax[nt];
            }
            __syncthreads();
            for (int o = tid; o < COUT; o += NTH) {
                float mx = red[o];
#pragma unroll
                for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w * COUT + o]);
                if (mx > a.absmax_out[o]) aq3d_atomic_max_pos(&a.absmax_out[o], mx);
            }
        }
    }
}

template <int CIN, int COUT, int KVOL, int NW>
static int launch_direct(const SpconvDirectArgs &a, hipStream_t stream) {
    constexpr int NCH = (KVOL * CIN + 31) / 32;
    constexpr size_t lds = (size_t)COUT * (NCH * 32 + 8) * 2 + NW * COUT * 4;
    const int tiles = aq3d_div_up(a.cap_out, 16 * NW);
    const int grid = tiles < 2048 ? (tiles < 1 ? 1 : tiles) : 2048;      // persistent: each workgroup stages the weights once
    auto kern = spconv_direct_kernel<CIN, COUT, KVOL, NW>;
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
    AQ3D_CHECK_LAUNCH("aq3d_spconv(direct gather)");
    return AQ3D_OK;
}

// AQ3D_ERR_UNSUPPORTED when the shape has no direct-gather variant (caller falls back to the LDS-staged kernel)
int aq3d_spconv_direct_dispatch(const void *feats_split, unsigned feats_bytes, const void *weight_f16, const float *w_dequant,
                                const float *bias, const float *post_scale, const float *post_shift, const float *residual,
                                const int *nbr, const int *count_out, int cap_out, int cin, int cout, int kvol, int relu,
                                float *feats_out, float *absmax_out, hipStream_t stream) {
    static const bool off = getenv("AQ3D_SPCONV_NO_DIRECT") != nullptr;
    if (off || kvol != 27) return AQ3D_ERR_UNSUPPORTED;
    SpconvDirectArgs a;
    a.feats = (const _Float16 *)feats_split; a.weight = (const _Float16 *)weight_f16; a.w_dequant = w_dequant; a.bias = bias;
    a.post_scale = post_scale; a.post_shift = post_shift; a.residual = residual; a.nbr = nbr; a.count_out = count_out;
    a.feats_out = feats_out; a.absmax_out = absmax_out; a.cap_out = cap_out; a.kvol = kvol; a.relu = relu;
    a.feats_bytes = feats_bytes;
    // 8-wave workgroups: measured on the W150k recipe 42 -> 33 us for 32->32 at 65 k rows (the dilated second level), but
    // 23 -> 30 us at 30 k rows and slower for Cin 16 — so only the 32->32 layers (they run on the densest level) use them
    static const int nw = getenv("AQ3D_DIRECT_WAVES") ? atoi(getenv("AQ3D_DIRECT_WAVES")) : 0;
    if (nw >= 8 || nw == 0) {
        if (cin == 32 && cout == 32) return launch_direct<32, 32, 27, 8>(a, stream);
    }
    if (nw >= 8) {
        if (cin == 16 && cout == 16) return launch_direct<16, 16, 27, 8>(a, stream);
        if (cin == 16 && cout == 32) return launch_direct<16, 32, 27, 8>(a, stream);
    }
    if (cin == 16 && cout == 16) return launch_direct<16, 16, 27, 4>(a, stream);
    if (cin == 16 && cout == 32) return launch_direct<16, 32, 27, 4>(a, stream);
    if (cin == 32 && cout == 32) return launch_direct<32, 32, 27, 4>(a, stream);
    return AQ3D_ERR_UNSUPPORTED;
}
