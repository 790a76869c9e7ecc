// Sparse→dense scatter and the per-frame SmoothQuant parameter preparation of the dense int8 convs.
//
// Reference semantics:
//   HeightCompression (pcdet/models/backbones_2d/map_to_bev/height_compression.py:19-23):
//       dense()[b, c, z, y, x] = feats[m, c];  view → channel c*D + z.   We keep NHWC: out[b, y, x, c*D + z].
//   SQConv2d.forward (quant/smoothquant.py:58-85):
//       X = unfold(x) [L, Cin*k*k];  act_scale_j = max_l |X[l,j]|;  w_scale_j = max_oc |W[oc,j]|
//       s_j = act_scale_j^alpha / w_scale_j^(1-alpha), s_j == 0 → 1;   W' = W * s;  X' = X / s
//       W' → per-out-channel fake quant;  X' → per-tensor fake quant;  Y = X'q @ W'q^T
//   The im2col matrix is never materialised here: column j = (c, kh, kw) of X is a shifted window of channel c,
//   so act_scale_j is the abs-max of x[c] over that window.  Windows of stride-1/2, pad<=1 3x3 (or 1x1) convs are
//   unions of 16 "pieces" (first / last / interior-even / interior-odd rows x the same for columns); one pass
//   over the input produces the 16 x Cin piece maxima, and the per-tensor amax of X' follows analytically:
//   max_l |X[l,j] / s_j| = act_scale_j / s_j (fp32 division is monotone), so no second pass is needed.
#include "common.h"
#include "sq_layout.h"

// ------------------------------------------------------------------ scatter
// (measured round 3: a (row, channel quad) thread mapping with 16-byte feature loads is no faster — 18 against 16 us: the kernel is bound
// by the half-filled lines it writes, channel = c * D + z puts the two z planes of a pixel in alternating floats — and computing the
// first BEV conv's 16 piece maxima here from the live rows instead of by a pass over the 36 MB canvas is SLOWER, 43 us against 16 + 16.5:
// every workgroup flushes ~1 200 maxima by atomicMax onto the same 4 096 words.)
__global__ __launch_bounds__(256) void scatter_dense_kernel(const float *__restrict__ feats,
                                                            const int *__restrict__ coords, const int *count, int cap,
                                                            int C, int D, int H, int W, int B, float *__restrict__ out) {
    int n = *count;
    if (n > cap) n = cap;
    const long long total = (long long)n * C;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        int m = (int)(t / C), c = (int)(t - (long long)m * C);
        const int4 q = reinterpret_cast<const int4 *>(coords)[m];  // b, z, y, x
        if (q.x < 0 || q.x >= B || q.y < 0 || q.y >= D || q.z < 0 || q.z >= H || q.w < 0 || q.w >= W) continue;
        out[(((size_t)q.x * H + q.z) * W + q.w) * ((size_t)C * D) + (size_t)c * D + q.y] = feats[t];
    }
}

extern "C" int aq3d_scatter_dense(const float *feats, const int32_t *coords, const int32_t *count, int32_t cap,
                                  int32_t C, const int32_t shape[3], int32_t batch_size, float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(feats && coords && count && out, "aq3d_scatter_dense: null argument");
    AQ3D_REQUIRE(C > 0 && shape[0] > 0 && shape[1] > 0 && shape[2] > 0 && batch_size > 0, "aq3d_scatter_dense: bad sizes");
    const size_t bytes = (size_t)batch_size * shape[1] * shape[2] * C * shape[0] * sizeof(float);
    aq3d_zero_async(out, bytes, stream);
    hipLaunchKernelGGL(scatter_dense_kernel, dim3(aq3d_grid((int64_t)cap * C, 256)), dim3(256), 0, stream, feats, coords,
                       count, cap, C, shape[0], shape[1], shape[2], batch_size, out);
    AQ3D_CHECK_LAUNCH("aq3d_scatter_dense");
    return AQ3D_OK;
}

// ------------------------------------------------------------------ piece abs-max
__device__ __forceinline__ int piece_class(int i, int n) { return i == 0 ? 0 : (i == n - 1 ? 1 : 2 + (i & 1)); }

// grid = (H, SEG). One block = one input row segment; thread = (channel quad, pixel lane).
__global__ __launch_bounds__(256) void piece_absmax_kernel(const float *__restrict__ x, int H, int W, int C, int ldc,
                                                           float *__restrict__ pieces) {
    extern __shared__ unsigned int red[];  // [4 col classes][C]
    const int h = blockIdx.x;
    const int c4n = C >> 2;
    const int lanes = 256 / c4n;  // pixel lanes per pass (C <= 1024)
    const int cq = threadIdx.x % c4n, pl = threadIdx.x / c4n;
    const int w_per = (W + gridDim.y - 1) / gridDim.y;
    const int w0 = blockIdx.y * w_per, w1 = min(W, w0 + w_per);
    for (int i = threadIdx.x; i < 4 * C; i += 256) red[i] = 0u;
    __syncthreads();
    float4 m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pl < lanes) {
        for (int w = w0 + pl; w < w1; w += lanes) {
            const float4 v = *reinterpret_cast<const float4 *>(x + ((size_t)h * W + w) * ldc + cq * 4);
            const int cc = piece_class(w, W);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (cc == k) {
                    m[k].x = fmaxf(m[k].x, fabsf(v.x));
                    m[k].y = fmaxf(m[k].y, fabsf(v.y));
                    m[k].z = fmaxf(m[k].z, fabsf(v.z));
                    m[k].w = fmaxf(m[k].w, fabsf(v.w));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicMax(&red[k * C + cq * 4 + 0], __float_as_uint(m[k].x));
            atomicMax(&red[k * C + cq * 4 + 1], __float_as_uint(m[k].y));
            atomicMax(&red[k * C + cq * 4 + 2], __float_as_uint(m[k].z));
            atomicMax(&red[k * C + cq * 4 + 3], __float_as_uint(m[k].w));
        }
    }
    __syncthreads();
    const int rc = piece_class(h, H);
    unsigned int *dst = reinterpret_cast<unsigned int *>(pieces) + (size_t)rc * 4 * C;
    for (int i = threadIdx.x; i < 4 * C; i += 256) {
        unsigned int v = red[i];
        if (v > dst[i]) atomicMax(&dst[i], v);
    }
}

extern "C" int aq3d_piece_absmax(const float *x, int32_t H, int32_t W, int32_t C, int32_t ldc, float *pieces,
                                 void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(x && pieces, "aq3d_piece_absmax: null argument");
    AQ3D_REQUIRE(H >= 3 && W >= 3, "aq3d_piece_absmax: H=%d W=%d must be >= 3", H, W);
    AQ3D_REQUIRE(C >= 4 && C <= 1024 && (C & 3) == 0 && (ldc & 3) == 0 && ldc >= C, "aq3d_piece_absmax: C=%d ldc=%d", C, ldc);
    const int seg = H >= 128 ? 4 : 8;
    hipLaunchKernelGGL(piece_absmax_kernel, dim3(H, seg), dim3(256), (size_t)4 * C * sizeof(unsigned int), stream, x, H, W,
                       C, ldc, pieces);
    AQ3D_CHECK_LAUNCH("aq3d_piece_absmax");
    return AQ3D_OK;
}

// ------------------------------------------------------------------ SmoothQuant parameter block
extern "C" size_t aq3d_sq_param_bytes(int32_t Cin, int32_t Cout, int32_t ksize) {
    return sq_layout(Cin, Cout, ksize).total;
}

struct TapMasks {
    unsigned short m[9];
};

// Inclusion of the 16 pieces in the sampling window of every tap; fails when a piece is only partly covered.
static int build_tap_masks(const aq3d_conv2d_geom *g, TapMasks *tm) {
    auto cls = [](int i, int n) { return i == 0 ? 0 : (i == n - 1 ? 1 : 2 + (i & 1)); };
    for (int kh = 0; kh < g->ksize; ++kh) {
        for (int kw = 0; kw < g->ksize; ++kw) {
            int rows_hit[4] = {0, 0, 0, 0}, rows_all[4] = {0, 0, 0, 0};
            int cols_hit[4] = {0, 0, 0, 0}, cols_all[4] = {0, 0, 0, 0};
            for (int r = 0; r < g->H; ++r) {
                int t = r - kh + g->pad;
                bool hit = t >= 0 && t % g->stride == 0 && t / g->stride < g->Ho;
                rows_all[cls(r, g->H)]++;
                rows_hit[cls(r, g->H)] += hit;
            }
            for (int c = 0; c < g->W; ++c) {
                int t = c - kw + g->pad;
                bool hit = t >= 0 && t % g->stride == 0 && t / g->stride < g->Wo;
                cols_all[cls(c, g->W)]++;
                cols_hit[cls(c, g->W)] += hit;
            }
            unsigned short mask = 0;
            for (int rc = 0; rc < 4; ++rc) {
                if (rows_hit[rc] != 0 && rows_hit[rc] != rows_all[rc]) return -1;
                for (int cc = 0; cc < 4; ++cc) {
                    if (cols_hit[cc] != 0 && cols_hit[cc] != cols_all[cc]) return -1;
                    if (rows_hit[rc] && cols_hit[cc]) mask |= (unsigned short)(1u << (rc * 4 + cc));
                }
            }
            tm->m[kh * g->ksize + kw] = mask;
        }
    }
    return 0;
}

__device__ __forceinline__ float block_max_1024(float v, float *sh) {
    v = aq3d_wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, sh[w]);
    __syncthreads();
    return r;
}

// One SmoothQuant preparation problem (a layer) of a grouped launch; blockIdx.y selects it.
struct SqOne {
    const float *pieces, *w_colmax, *weight, *static_amax_w;
    float *s, *r, *amax_x_acc, *scal, *out_mul, *amax_w_out;
    int8_t *wq;
    int Cout;
    float alpha, static_amax_x;
};
#define AQ3D_SQ_MAX_GROUP 8
struct SqGroup {
    SqOne p[AQ3D_SQ_MAX_GROUP];
};

// smoothing scales, one column per thread; the per-tensor activation amax max_j a_j / s_j is accumulated with one atomic
// per block into a caller-zeroed slot
__global__ __launch_bounds__(256) void sq_scales_kernel(SqGroup grp, int Cin, int ntaps, TapMasks tm) {
    const SqOne q = grp.p[blockIdx.y];
    __shared__ float sh[16];
    const int K = Cin * ntaps;
    const int j = blockIdx.x * 256 + threadIdx.x;
    float t = 0.f;
    if (j < K) {
        const int tap = j / Cin, c = j - tap * Cin;
        const unsigned mask = tm.m[tap];
        float a = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p)
            if (mask & (1u << p)) a = fmaxf(a, q.pieces[p * Cin + c]);
        float s = 1.f;
        if (q.alpha >= 0.f) {
            const float w = q.w_colmax[j];
            // torch.pow(x, 0.5) takes the sqrt path; other exponents go through powf
            if (q.alpha == 0.5f) s = __fdiv_rn(sqrtf(a), sqrtf(w));
            else s = __fdiv_rn(powf(a, q.alpha), powf(w, 1.f - q.alpha));
            if (s == 0.f) s = 1.f;
        }
        q.s[j] = s;
        q.r[j] = __fdiv_rn(1.0f, s);
        t = fabsf(__fdiv_rn(a, s));
    }
    const float m = block_max_1024(t, sh);
    if (threadIdx.x == 0 && m > 0.f && m > *q.amax_x_acc) aq3d_atomic_max_pos(q.amax_x_acc, m);
}

// one block per output channel: W' = W * s, per-row amax, int8 codes, combined dequant multiplier
__global__ __launch_bounds__(256) void sq_weights_kernel(SqGroup grp, int K, float qmax_w, float qmax_a) {
    const SqOne q = grp.p[blockIdx.y];
    const int oc = blockIdx.x;
    if (oc >= q.Cout) return;
    __shared__ float sh[16];
    const float eps = 1.0f / (float)(1 << 24);
    // per-tensor activation scale (every block derives it; block 0 publishes it for the conv kernel)
    const float amax_x = q.static_amax_x > 0.f ? q.static_amax_x : *q.amax_x_acc;
    float sx = __fdiv_rn(qmax_a, amax_x), dvx = sx;
    if (amax_x <= eps) { sx = 0.f; dvx = 1.f; }
    const float inv_sx = __fdiv_rn(1.0f, dvx);
    if (oc == 0 && threadIdx.x == 0) { q.scal[0] = sx; q.scal[1] = amax_x; q.scal[2] = inv_sx; q.scal[3] = q.static_amax_x > 0.f ? 1.f : 0.f; q.scal[4] = qmax_a; }
    const float *wrow = q.weight + (size_t)oc * K;
    float mx = 0.f;
    for (int j = threadIdx.x; j < K; j += 256) mx = fmaxf(mx, fabsf(__fmul_rn(wrow[j], q.s[j])));
    float amax = block_max_1024(mx, sh);
    if (threadIdx.x == 0) q.amax_w_out[oc] = amax;   // dynamic per-oc amax of W*s (what a max calibrator collects)
    if (q.static_amax_w) amax = q.static_amax_w[oc];
    float sw = __fdiv_rn(qmax_w, amax), dv = sw;
    if (amax <= eps) { sw = 0.f; dv = 1.f; }
    for (int j = threadIdx.x; j < K; j += 256) {
        float v = rintf(__fmul_rn(__fmul_rn(wrow[j], q.s[j]), sw));
        v = fminf(fmaxf(v, -qmax_w - 1.f), qmax_w);
        q.wq[(size_t)oc * K + j] = (int8_t)(int)v;
    }
    if (threadIdx.x == 0) q.out_mul[oc] = __fmul_rn(inv_sx, __fdiv_rn(1.0f, dv));
}

// Fused form of the two kernels above: ONE launch per layer group instead of two dependent ones (each dependent launch in a
// captured frame costs its run time plus ~6 us of dispatch gap, 14 layers per frame). One block per output channel, as in
// sq_weights_kernel. A thread owns up to SQ_V groups of four consecutive columns (same tap, Cin % 4 == 0): its weight and
// w_colmax quads are requested first, so that their latency overlaps the one coalesced round trip that stages the 16 x Cin
// piece maxima in LDS; the smoothing scales of its columns are then computed in registers (redundantly across blocks, but
// concurrently), and the weight row is quantised from registers — one global round trip in all. Block x == 0 publishes s, 1/s,
// the activation scale and the amax. Same arithmetic, same results as the two-kernel path.
#define SQ_FUSED_THREADS 512
#define SQ_V 3
__device__ __forceinline__ float sq_scale_of(float a, float w, float alpha) {
    float s = 1.f;
    if (alpha >= 0.f) {
        // torch.pow(x, 0.5) takes the sqrt path; other exponents go through powf
        if (alpha == 0.5f) s = __fdiv_rn(sqrtf(a), sqrtf(w));
        else s = __fdiv_rn(powf(a, alpha), powf(w, 1.f - alpha));
        if (s == 0.f) s = 1.f;
    }
    return s;
}
__device__ __forceinline__ signed char sq_code(float w, float s, float sw, float qmax) {
    return (signed char)(int)fminf(fmaxf(rintf(__fmul_rn(__fmul_rn(w, s), sw)), -qmax - 1.f), qmax);
}
__global__ __launch_bounds__(SQ_FUSED_THREADS) void sq_prep_row_kernel(SqGroup grp, int Cin, int ntaps, TapMasks tm, float qmax_w,
                                                                       float qmax_a) {
    const SqOne q = grp.p[blockIdx.y];
    const int oc = blockIdx.x;
    if (oc >= q.Cout) return;
    extern __shared__ float4 p_lds[];       // [16][Cin/4]
    __shared__ float sh[16];
    const int K4 = (Cin * ntaps) >> 2, C4 = Cin >> 2;
    const float eps = 1.0f / (float)(1 << 24);
    const bool pub = oc == 0;
    const float4 *wrow = reinterpret_cast<const float4 *>(q.weight + (size_t)oc * K4 * 4);
    const float4 *wcm = reinterpret_cast<const float4 *>(q.w_colmax);
    float4 w[SQ_V], wc[SQ_V], sv[SQ_V];
#pragma unroll
    for (int i = 0; i < SQ_V; ++i) {
        const int g = threadIdx.x + i * SQ_FUSED_THREADS;
        w[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        wc[i] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (g < K4) {
            w[i] = wrow[g];
            if (q.alpha >= 0.f) wc[i] = wcm[g];
        }
    }
    for (int i = threadIdx.x; i < 16 * C4; i += SQ_FUSED_THREADS) p_lds[i] = reinterpret_cast<const float4 *>(q.pieces)[i];
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < SQ_V; ++i) {
        const int g = threadIdx.x + i * SQ_FUSED_THREADS;
        sv[i] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (g < K4) {
            const int tap = g / C4, c4 = g - tap * C4;
            const unsigned mask = tm.m[tap];
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (mask & (1u << p)) {     // uniform per tap; at most one tap boundary inside a wave
                    const float4 v = p_lds[p * C4 + c4];
                    a.x = fmaxf(a.x, v.x); a.y = fmaxf(a.y, v.y); a.z = fmaxf(a.z, v.z); a.w = fmaxf(a.w, v.w);
                }
            }
            float4 sc;
            sc.x = sq_scale_of(a.x, wc[i].x, q.alpha); sc.y = sq_scale_of(a.y, wc[i].y, q.alpha);
            sc.z = sq_scale_of(a.z, wc[i].z, q.alpha); sc.w = sq_scale_of(a.w, wc[i].w, q.alpha);
            sv[i] = sc;
            if (pub) {
                reinterpret_cast<float4 *>(q.s)[g] = sc;
                reinterpret_cast<float4 *>(q.r)[g] = make_float4(__fdiv_rn(1.0f, sc.x), __fdiv_rn(1.0f, sc.y), __fdiv_rn(1.0f, sc.z),
                                                                 __fdiv_rn(1.0f, sc.w));
            }
            t = fmaxf(fmaxf(t, fabsf(__fdiv_rn(a.x, sc.x))), fabsf(__fdiv_rn(a.y, sc.y)));
            t = fmaxf(fmaxf(t, fabsf(__fdiv_rn(a.z, sc.z))), fabsf(__fdiv_rn(a.w, sc.w)));
        }
    }
    const float amax_dyn = block_max_1024(t, sh);
    const float amax_x = q.static_amax_x > 0.f ? q.static_amax_x : amax_dyn;
    float sx = __fdiv_rn(qmax_a, amax_x), dvx = sx;
    if (amax_x <= eps) { sx = 0.f; dvx = 1.f; }
    const float inv_sx = __fdiv_rn(1.0f, dvx);
    if (pub && threadIdx.x == 0) {
        q.scal[0] = sx; q.scal[1] = amax_x; q.scal[2] = inv_sx; q.scal[3] = q.static_amax_x > 0.f ? 1.f : 0.f; q.scal[4] = qmax_a;
        if (amax_dyn > 0.f && amax_dyn > *q.amax_x_acc) aq3d_atomic_max_pos(q.amax_x_acc, amax_dyn);
    }
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < SQ_V; ++i) {        // absent groups hold w = 0
        mx = fmaxf(fmaxf(mx, fabsf(__fmul_rn(w[i].x, sv[i].x))), fabsf(__fmul_rn(w[i].y, sv[i].y)));
        mx = fmaxf(fmaxf(mx, fabsf(__fmul_rn(w[i].z, sv[i].z))), fabsf(__fmul_rn(w[i].w, sv[i].w)));
    }
    float amax = block_max_1024(mx, sh);
    if (threadIdx.x == 0) q.amax_w_out[oc] = amax;   // dynamic per-oc amax of W*s (what a max calibrator collects)
    if (q.static_amax_w) amax = q.static_amax_w[oc];
    float sw = __fdiv_rn(qmax_w, amax), dv = sw;
    if (amax <= eps) { sw = 0.f; dv = 1.f; }
    char4 *dst = reinterpret_cast<char4 *>(q.wq + (size_t)oc * K4 * 4);
#pragma unroll
    for (int i = 0; i < SQ_V; ++i) {
        const int g = threadIdx.x + i * SQ_FUSED_THREADS;
        if (g < K4)
            dst[g] = make_char4(sq_code(w[i].x, sv[i].x, sw, qmax_w), sq_code(w[i].y, sv[i].y, sw, qmax_w),
                                sq_code(w[i].z, sv[i].z, sw, qmax_w), sq_code(w[i].w, sv[i].w, sw, qmax_w));
    }
    if (threadIdx.x == 0) q.out_mul[oc] = __fmul_rn(inv_sx, __fdiv_rn(1.0f, dv));
}

extern "C" int aq3d_sq_prep_group(const aq3d_sq_prep_problem *probs, int32_t n, int32_t w_bits, int32_t a_bits, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ3D_REQUIRE(probs && n >= 1 && n <= AQ3D_SQ_MAX_GROUP, "aq3d_sq_prep_group: n=%d not in 1..%d", n, AQ3D_SQ_MAX_GROUP);
    AQ3D_REQUIRE(w_bits >= 2 && w_bits <= 8 && a_bits >= 2 && a_bits <= 16, "aq3d_sq_prep: w_bits <= 8, a_bits <= 16 (w=%d a=%d)", w_bits, a_bits);
    const aq3d_conv2d_geom *g = probs[0].geom;
    AQ3D_REQUIRE(g, "aq3d_sq_prep: null argument");
    AQ3D_REQUIRE(g->ksize == 1 || g->ksize == 3, "aq3d_sq_prep: ksize=%d", g->ksize);
    AQ3D_REQUIRE(g->stride >= 1 && g->pad >= 0 && g->H >= 3 && g->W >= 3, "aq3d_sq_prep: bad geometry");
    TapMasks tm;
    if (build_tap_masks(g, &tm) != 0) {
        aq3d_set_error("aq3d_sq_prep: a tap window covers a border piece only partly (H=%d W=%d k=%d s=%d p=%d)", g->H,
                       g->W, g->ksize, g->stride, g->pad);
        return AQ3D_ERR_UNSUPPORTED;
    }
    SqGroup grp;
    int max_cout = 0;
    for (int i = 0; i < n; ++i) {
        const aq3d_sq_prep_problem &q = probs[i];
        AQ3D_REQUIRE(q.pieces && q.weight && q.geom && q.params && q.amax_x_acc, "aq3d_sq_prep: null argument");
        AQ3D_REQUIRE(q.alpha < 0.f || q.w_colmax, "aq3d_sq_prep: w_colmax needed when smoothing is on");
        AQ3D_REQUIRE(q.geom->H == g->H && q.geom->W == g->W && q.geom->Cin == g->Cin && q.geom->ksize == g->ksize &&
                     q.geom->stride == g->stride && q.geom->pad == g->pad, "aq3d_sq_prep_group: problem %d has another geometry", i);
        const SqLayout l = sq_layout(g->Cin, q.geom->Cout, g->ksize);
        char *base = (char *)q.params;
        SqOne &o = grp.p[i];
        o.pieces = q.pieces; o.w_colmax = q.w_colmax; o.weight = q.weight; o.static_amax_w = q.static_amax_w;
        o.s = (float *)(base + l.s); o.r = (float *)(base + l.r); o.out_mul = (float *)(base + l.mul);
        o.scal = (float *)(base + l.scal); o.amax_w_out = (float *)(base + l.amaxw); o.wq = (int8_t *)(base + l.wq);
        o.amax_x_acc = q.amax_x_acc; o.Cout = q.geom->Cout; o.alpha = q.alpha; o.static_amax_x = q.static_amax_x;
        max_cout = o.Cout > max_cout ? o.Cout : max_cout;
    }
    for (int i = n; i < AQ3D_SQ_MAX_GROUP; ++i) grp.p[i] = grp.p[0];
    const int ntaps = g->ksize * g->ksize, K = g->Cin * ntaps;
    const float qa = (float)((1 << (a_bits - 1)) - 1), qw = (float)((1 << (w_bits - 1)) - 1);
    // one launch (scales and codes from registers, piece maxima through LDS) when a weight row fits the block; otherwise two
    // kernels (scales, then one block per output channel)
    const size_t lds_bytes = 16 * (size_t)g->Cin * 4;
    if ((g->Cin & 3) == 0 && K <= 4 * SQ_V * SQ_FUSED_THREADS && lds_bytes <= 60 * 1024) {
        hipLaunchKernelGGL(sq_prep_row_kernel, dim3(max_cout, n), dim3(SQ_FUSED_THREADS), lds_bytes, stream, grp, g->Cin, ntaps, tm,
                           qw, qa);
        AQ3D_CHECK_LAUNCH("aq3d_sq_prep");
        return AQ3D_OK;
    }
    hipLaunchKernelGGL(sq_scales_kernel, dim3(aq3d_div_up(K, 256), n), dim3(256), 0, stream, grp, g->Cin, ntaps, tm);
    hipLaunchKernelGGL(sq_weights_kernel, dim3(max_cout, n), dim3(256), 0, stream, grp, K, qw, qa);
    AQ3D_CHECK_LAUNCH("aq3d_sq_prep");
    return AQ3D_OK;
}

extern "C" int aq3d_sq_prep(const float *pieces, const float *weight, const float *w_colmax,
                            const aq3d_conv2d_geom *g, float alpha, int32_t w_bits, int32_t a_bits,
                            float static_amax_x, const float *static_amax_w, float *amax_x_acc, void *params,
                            void *stream_) {
    aq3d_sq_prep_problem q;
    q.pieces = pieces; q.weight = weight; q.w_colmax = w_colmax; q.geom = g; q.alpha = alpha; q.static_amax_x = static_amax_x;
    q.static_amax_w = static_amax_w; q.amax_x_acc = amax_x_acc; q.params = params;
    return aq3d_sq_prep_group(&q, 1, w_bits, a_bits, stream_);
}
