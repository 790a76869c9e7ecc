/* ORACLE (test infrastructure): rotated BEV IoU + greedy NMS in plain C.
 * Restates pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:59-229 (box_overlap / iou_bev geometry, EPS 1e-8, MARGIN 1e-2)
 * and the host sweep of pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:163-179 (boxes pre-sorted by score, keep order).
 * The reference file itself needs <cuda.h> / torch headers and is not buildable in this image. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } pt;
static const float EPS = 1e-8f;

static float crs3(pt p1, pt p2, pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
static float crs2(pt a, pt b) { return a.x * b.y - a.y * b.x; }
static float fmn(float a, float b) { return a > b ? b : a; }
static float fmx(float a, float b) { return a > b ? a : b; }

static int rect_cross(pt p1, pt p2, pt q1, pt q2) {
    return fmn(p1.x, p2.x) <= fmx(q1.x, q2.x) && fmn(q1.x, q2.x) <= fmx(p1.x, p2.x) &&
           fmn(p1.y, p2.y) <= fmx(q1.y, q2.y) && fmn(q1.y, q2.y) <= fmx(p1.y, p2.y);
}

static int in_box(const float *box, pt p) {
    const float MARGIN = 1e-2f;
    float ac = cosf(-box[6]), as = sinf(-box[6]);
    float rx = (p.x - box[0]) * ac + (p.y - box[1]) * (-as);
    float ry = (p.x - box[0]) * as + (p.y - box[1]) * ac;
    return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}

static int isect(pt p1, pt p0, pt q1, pt q0, pt *ans) {
    if (!rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = crs3(q0, p1, p0), s2 = crs3(p1, q1, p0), s3 = crs3(p0, q1, q0), s4 = crs3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = crs3(q1, p1, p0);
    if (fabsf(s5 - s1) > EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

static void corners(const float *b, pt *c) {
    float hx = b[3] / 2, hy = b[4] / 2;
    float xs[4] = {b[0] - hx, b[0] + hx, b[0] + hx, b[0] - hx};
    float ys[4] = {b[1] - hy, b[1] - hy, b[1] + hy, b[1] + hy};
    float ac = cosf(b[6]), as = sinf(b[6]);
    for (int k = 0; k < 4; k++) {
        c[k].x = (xs[k] - b[0]) * ac + (ys[k] - b[1]) * (-as) + b[0];
        c[k].y = (xs[k] - b[0]) * as + (ys[k] - b[1]) * ac + b[1];
    }
    c[4] = c[0];
}

float oracle_box_overlap(const float *ba, const float *bb) {
    pt ca[5], cb[5], pts[16], ctr = {0.f, 0.f};
    int cnt = 0;
    corners(ba, ca);
    corners(bb, cb);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            pt ip;
            if (isect(ca[i + 1], ca[i], cb[j + 1], cb[j], &ip)) { ctr.x += ip.x; ctr.y += ip.y; pts[cnt++] = ip; }
        }
    for (int k = 0; k < 4; k++) {
        if (in_box(ba, cb[k])) { ctr.x += cb[k].x; ctr.y += cb[k].y; pts[cnt++] = cb[k]; }
        if (in_box(bb, ca[k])) { ctr.x += ca[k].x; ctr.y += ca[k].y; pts[cnt++] = ca[k]; }
    }
    ctr.x /= cnt;
    ctr.y /= cnt;
    for (int j = 0; j < cnt - 1; j++)
        for (int i = 0; i < cnt - j - 1; i++)
            if (atan2f(pts[i].y - ctr.y, pts[i].x - ctr.x) > atan2f(pts[i + 1].y - ctr.y, pts[i + 1].x - ctr.x)) {
                pt t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
            }
    float area = 0;
    for (int k = 0; k < cnt - 1; k++) {
        pt u = {pts[k].x - pts[0].x, pts[k].y - pts[0].y}, v = {pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y};
        area += crs2(u, v);
    }
    return fabsf(area) / 2.0f;
}

float oracle_iou_bev(const float *a, const float *b) {
    float sa = a[3] * a[4], sb = b[3] * b[4];
    float ov = oracle_box_overlap(a, b);
    return ov / fmaxf(sa + sb - ov, EPS);
}

/* overlap[n, m] (rotated BEV intersection area) between boxes_a [n, lda] and boxes_b [m, ldb]. */
void oracle_overlap_bev_matrix(const float *a, int n, int lda, const float *b, int m, int ldb, float *ov) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) ov[(size_t)i * m + j] = oracle_box_overlap(a + (size_t)i * lda, b + (size_t)j * ldb);
}

/* iou[n, m] between boxes_a [n, lda] and boxes_b [m, ldb] (first 7 columns used). */
void oracle_iou_bev_matrix(const float *a, int n, int lda, const float *b, int m, int ldb, float *iou) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) iou[(size_t)i * m + j] = oracle_iou_bev(a + (size_t)i * lda, b + (size_t)j * ldb);
}

/* boxes sorted by descending score; returns number kept, indices in keep[]. */
int oracle_nms(const float *boxes, int n, int ld, float thresh, int *keep) {
    unsigned char *removed = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int num = 0;
    for (int i = 0; i < n; i++) {
        if (removed[i]) continue;
        keep[num++] = i;
        for (int j = i + 1; j < n; j++)
            if (!removed[j] && oracle_iou_bev(boxes + (size_t)i * ld, boxes + (size_t)j * ld) > thresh) removed[j] = 1;
    }
    free(removed);
    return num;
}
