// Layout of the per-layer, per-frame SmoothQuant parameter block (written by aq3d_sq_prep, read by aq3d_conv2d_i8).
//   s f32[K] | rinv f32[K] | out_mul f32[Cout] | amax_w f32[Cout] | scal f32[64] | wq i8[Cout*K]     (256-B aligned parts)
//   scal[0] = activation quant scale sx (0 if amax_x <= 2^-24), scal[1] = amax_x, scal[2] = 1/sx, scal[3] = static-amax flag,
//   scal[4] = qmax of the activation quantiser (127, or 32767 for the 16-bit path)
#pragma once
#include <stddef.h>

struct SqLayout {
    size_t s, r, mul, amaxw, scal, wq, total;
};

static inline size_t sq_align(size_t v) { return (v + 255) / 256 * 256; }

static inline SqLayout sq_layout(int Cin, int Cout, int ksize) {
    const size_t K = (size_t)Cin * ksize * ksize;
    SqLayout l;
    l.s = 0;
    l.r = l.s + sq_align(K * 4);
    l.mul = l.r + sq_align(K * 4);
    l.amaxw = l.mul + sq_align((size_t)Cout * 4);
    l.scal = l.amaxw + sq_align((size_t)Cout * 4);
    l.wq = l.scal + 256;
    l.total = l.wq + sq_align((size_t)Cout * K);
    return l;
}
