// Rotated BEV overlap of two boxes — device functions shared by the NMS mask kernel, the 3-D IoU kernel (decode_nms.hip)
// and the diagnostic build (csrc/diag/nms_diag.hip, never linked into libaq3d.so).
//
// Convex clipping of two rotated rectangles by collecting edge-edge intersections and contained corners, ordering them by
// angle around their centroid and summing the fan area — the reference's algorithm, tolerances included (EPS 1e-8 on the
// parallel test, 1e-2 margin on the corner-containment test): pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:35-234.
#pragma once
#include <hip/hip_runtime.h>

struct P2 { float x, y; };
#define IOU_EPS 1e-8f
#define IOU_MARGIN 1e-2f
// 16 edge-edge crossings + 8 contained corners (the 1 cm containment margin lets a corner count twice as "inside"): the
// reference sizes its array for 16 and overruns it silently; here the polygon has room for all 24 candidates.
#define NMS_MAX_PTS 24

__device__ __forceinline__ float cross3(const P2 &p1, const P2 &p2, const P2 &p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
__device__ __forceinline__ float cross2(const P2 &a, const P2 &b) { return a.x * b.y - a.y * b.x; }

__device__ __forceinline__ bool bbox_overlap(const P2 &p1, const P2 &p2, const P2 &q1, const P2 &q2) {
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ __forceinline__ bool seg_intersection(const P2 &p1, const P2 &p0, const P2 &q1, const P2 &q0, P2 &ans) {
    if (!bbox_overlap(p0, p1, q0, q1)) return false;
    const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0.f && s3 * s4 > 0.f)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__device__ __forceinline__ bool corner_in_box(const float *box, const P2 &p) {
    const float ac = cosf(-box[6]), as = sinf(-box[6]);
    const float rx = (p.x - box[0]) * ac + (p.y - box[1]) * (-as);
    const float ry = (p.x - box[0]) * as + (p.y - box[1]) * ac;
    return fabsf(rx) < box[3] / 2 + IOU_MARGIN && fabsf(ry) < box[4] / 2 + IOU_MARGIN;
}

__device__ __forceinline__ void box_corners(const float *b, P2 *c) {
    const float hx = b[3] / 2, hy = b[4] / 2;
    const float x1 = b[0] - hx, y1 = b[1] - hy, x2 = b[0] + hx, y2 = b[1] + hy;
    const float ac = cosf(b[6]), as = sinf(b[6]);
    const float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k].x = (px[k] - b[0]) * ac + (py[k] - b[1]) * (-as) + b[0];
        c[k].y = (px[k] - b[0]) * as + (py[k] - b[1]) * ac + b[1];
    }
    c[4] = c[0];
}

// The intersection polygon (<= NMS_MAX_PTS points, indexed at run time) lives in LDS: pts_lds points at this thread's first
// element, consecutive points are PTS_STRIDE apart (a per-thread array would be scratch memory, the only scratch of the frame).
template <int PTS_STRIDE>
__device__ float rotated_overlap(const float *ba, const float *bb, P2 *pts_lds, int *cnt_out = nullptr) {
    P2 ca[5], cb[5];
    struct Pts {
        P2 *base;
        __device__ __forceinline__ P2 &operator[](int k) const { return base[k * PTS_STRIDE]; }
    } pts = {pts_lds};
    box_corners(ba, ca);
    box_corners(bb, cb);
    int cnt = 0;
    P2 ctr = {0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 ip;
            if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], ip)) {
                ctr.x += ip.x; ctr.y += ip.y;
                pts[cnt++] = ip;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (corner_in_box(ba, cb[k])) { ctr.x += cb[k].x; ctr.y += cb[k].y; pts[cnt++] = cb[k]; }
        if (corner_in_box(bb, ca[k])) { ctr.x += ca[k].x; ctr.y += ca[k].y; pts[cnt++] = ca[k]; }
    }
    ctr.x /= cnt;
    ctr.y /= cnt;
    // bubble sort by polar angle around the centroid (same comparison as the reference: strict >)
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (atan2f(pts[i].y - ctr.y, pts[i].x - ctr.x) > atan2f(pts[i + 1].y - ctr.y, pts[i + 1].x - ctr.x)) {
                P2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        P2 u = {pts[k].x - pts[0].x, pts[k].y - pts[0].y}, v = {pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y};
        area += cross2(u, v);
    }
    if (cnt_out) *cnt_out = cnt;
    return fabsf(area) / 2.0f;
}

template <int PTS_STRIDE>
__device__ __forceinline__ float iou_bev_dev(const float *a, const float *b, P2 *pts_lds) {
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    // The same BEV rectangle twice (one pixel ranked in two classes: the regression maps are shared) is the degenerate case of
    // the clipping — eight coincident corner pairs ordered by angle, fan area decided by last-bit ties. The exact overlap of
    // a rectangle with itself is its area.
    const bool same = a[0] == b[0] && a[1] == b[1] && a[3] == b[3] && a[4] == b[4] && a[6] == b[6];
    const float ov = same ? sa : rotated_overlap<PTS_STRIDE>(a, b, pts_lds);
    return ov / fmaxf(sa + sb - ov, IOU_EPS);
}

// Circumscribed circles more than 2 cm apart: no edge can cross and no corner can pass the 1 cm containment margin, so the
// clipping result is exactly 0 — an optimisation that decides most of the 125 k pairs of a 500-box list without running it.
__device__ __forceinline__ bool circles_may_touch(const float *a, const float *b) {
    const float ddx = a[0] - b[0], ddy = a[1] - b[1];
    const float reach = 0.5f * (sqrtf(a[3] * a[3] + a[4] * a[4]) + sqrtf(b[3] * b[3] + b[4] * b[4])) + 0.02f;
    return ddx * ddx + ddy * ddy <= reach * reach;
}
