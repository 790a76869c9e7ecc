"""GPU parity: sparse→dense scatter, SmoothQuant int8 conv, fp32 conv / deconv, decode and rotated NMS vs the oracle.

Bars: scatter exact; int8 conv within 2e-5 of the output scale (int32 accumulation is exact, the reference's fp32
matmul is not — SURVEY.md §8a row 13 measured 3e-6); fp32 convs within 2e-5; decode scores/boxes within 1e-5 with
identical ranking; NMS keep lists identical.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import decode as OD
from oracle import dense as ODN

pytestmark = pytest.mark.gpu


def _nhwc(t):   # [1,C,H,W] → [H,W,C]
    return t[0].permute(1, 2, 0).contiguous()


def test_scatter_dense_exact(dev):
    from amdq3d import ops
    rng = np.random.default_rng(0)
    D, H, W, C, B = 2, 24, 20, 128, 2
    idx = np.unique(np.stack([rng.integers(0, B, 700), rng.integers(0, D, 700), rng.integers(0, H, 700),
                              rng.integers(0, W, 700)], 1).astype(np.int32), axis=0)
    rng.shuffle(idx)
    m = len(idx)
    feats = rng.normal(0, 1, (m, C)).astype(np.float32)
    ref = ODN.height_compression(feats, idx, (D, H, W), B)          # [B, C*D, H, W]
    cap = m + 9
    fc = torch.zeros((cap, C), device=dev); fc[:m] = torch.from_numpy(feats).to(dev)
    cc = torch.zeros((cap, 4), dtype=torch.int32, device=dev); cc[:m] = torch.from_numpy(idx).to(dev)
    cnt = torch.tensor([m], dtype=torch.int32, device=dev)
    out = ops.scatter_dense(fc, cc, cnt, (D, H, W), B).cpu()          # [B, H, W, C*D]
    np.testing.assert_array_equal(out.permute(0, 3, 1, 2).numpy(), ref.numpy())


CASES_I8 = [
    # Cin, Cout, H, W, stride, alpha, bias, bn
    (64, 64, 24, 20, 1, 0.5, True, True),
    (128, 128, 40, 36, 1, 0.5, False, True),
    (128, 256, 36, 40, 2, 0.5, False, True),
    (256, 128, 21, 19, 1, 0.5, False, False),
    (64, 64, 16, 16, 1, None, True, False),
    (512, 64, 20, 24, 1, 0.3, True, True),
    (128, 128, 70, 45, 1, 0.5, False, True),        # several pixel tiles in both directions, ragged right / bottom edges
    (256, 256, 47, 47, 1, 0.5, True, True),         # a 256-channel tile (8 waves' worth of rows), odd image size
    (128, 200, 23, 37, 2, 0.5, False, True),        # Cout that is no multiple of 64: the rows past Cout are computed, never stored
    (128, 128, 188, 188, 1, 0.5, False, True),      # the benchmark's BEV layer: 64-pixel tiles, four pixels per thread and image row (row-register kernel)
    (64, 64, 186, 180, 1, 0.5, True, True),         # 64-pixel tiles, 64 output channels, a tile count that is no multiple of 8
    (128, 128, 182, 182, 1, 0.5, False, True),      # 64-pixel tiles on a width that is no multiple of 4: the per-step kernel
    (256, 256, 94, 94, 1, 0.5, False, True),        # 32-pixel tiles, two pixels per thread and row, two output-channel blocks
]


@pytest.mark.parametrize("Cin,Cout,H,W,stride,alpha,use_bias,use_bn", CASES_I8)
def test_conv2d_i8_vs_sqconv2d(dev, Cin, Cout, H, W, stride, alpha, use_bias, use_bn):
    from amdq3d import ops
    g_ = torch.Generator().manual_seed(Cin * 3 + Cout + H)
    x = torch.relu(torch.randn(1, Cin, H, W, generator=g_)) * (torch.rand(1, Cin, 1, 1, generator=g_) * 3 + 0.1)
    x[:, :, 0, :] *= 2.5      # make border pieces matter for the column abs-max
    x[:, :, :, -1] *= 1.7
    w = torch.randn(Cout, Cin, 3, 3, generator=g_) / np.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g_) * 0.1 if use_bias else None
    ref, st = ODN.sq_conv2d(x, w, bias, stride, 1, alpha, return_stats=True)
    ps = pt = None
    if use_bn:
        ps = torch.rand(Cout, generator=g_) + 0.5
        pt = torch.randn(Cout, generator=g_) * 0.1
        ref = torch.relu(ref * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))
    xg = _nhwc(x).to(dev)
    geom = ops.conv_geom(H, W, Cin, Cout, 3, stride, 1, relu=use_bn)
    wk = ops.conv_weight_to_k(w).to(dev)
    wcol = wk.abs().max(dim=0)[0].contiguous()
    pieces = ops.piece_absmax(xg, H, W, Cin)
    params = ops.sq_prep(pieces, wk, wcol, geom, alpha)
    po = torch.zeros((16, Cout), device=dev)
    kw = dict(bias=None if bias is None else bias.to(dev), post_scale=None if ps is None else ps.to(dev),
              post_shift=None if pt is None else pt.to(dev))
    y = ops.conv2d_i8(xg, params, geom, pieces_out=po, **kw)
    # fused piece maxima of the output == the standalone pass over the output
    np.testing.assert_array_equal(po.cpu().numpy(), ops.piece_absmax(y, geom.Ho, geom.Wo, Cout).cpu().numpy())
    # the merged-interior piece mode: same output bits, its interior classes' maximum is the interior abs-max, the 12 border classes exact
    po2 = torch.zeros((16, Cout), device=dev)
    y2 = ops.conv2d_i8(xg, params, geom, pieces_out=po2, pieces_mode=1, **kw)
    assert torch.equal(y2, y)
    a_, b_ = po2.cpu().numpy(), po.cpu().numpy()
    inter = [10, 11, 14, 15]
    border = [i for i in range(16) if i not in inter]
    np.testing.assert_array_equal(a_[border], b_[border])
    np.testing.assert_array_equal(a_[inter].max(axis=0), b_[inter].max(axis=0))
    y = y.cpu().permute(2, 0, 1)[None]
    amax_x = ops.sq_param_views(params, Cin, Cout, 3)["scal"][1].item()
    assert amax_x == pytest.approx(float(st["amax_x"]), rel=1e-6)
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert float((y - ref).abs().max()) <= tol


@pytest.mark.parametrize("H,W", [(188, 188), (186, 186)])
def test_conv2d_i8_group_equals_single_launches(dev, H, W):
    """The head's first convs: five 64 -> 64 SQConv2d problems of one launch class in ONE launch (aq3d_conv2d_i8_group). At 188 x 188 the
    group runs the row-register kernel on 64-pixel tiles (where the tile rule alone would pick 128-pixel tiles), at 186 x 186 — a width
    that is no multiple of 4 — the per-step kernel on 128-pixel tiles; every problem must equal its single launch bit for bit."""
    from amdq3d import ops
    g_ = torch.Generator().manual_seed(H)
    Cin = Cout = 64
    x = (torch.relu(torch.randn(H, W, Cin, generator=g_)) * (torch.rand(Cin, generator=g_) * 3 + 0.1)).to(dev)
    geom = ops.conv_geom(H, W, Cin, Cout, 3, 1, 1, relu=True)
    pieces = ops.piece_absmax(x, H, W, Cin)
    recs, singles = [], []
    for k in range(5):
        w = torch.randn(Cout, Cin, 3, 3, generator=g_) / np.sqrt(9 * Cin)
        wk = ops.conv_weight_to_k(w).to(dev)
        params = ops.sq_prep(pieces, wk, wk.abs().max(dim=0)[0].contiguous(), geom, 0.5)
        bias = (torch.randn(Cout, generator=g_) * 0.1).to(dev)
        ps, pt = (torch.rand(Cout, generator=g_) + 0.5).to(dev), (torch.randn(Cout, generator=g_) * 0.1).to(dev)
        singles.append(ops.conv2d_i8(x, params, geom, bias=bias, post_scale=ps, post_shift=pt))
        recs.append(dict(x=x, w=params, g=geom, bias=bias, post_scale=ps, post_shift=pt, y=torch.zeros((H, W, Cout), device=dev)))
    ops.conv2d_i8_group(ops.conv_problems(recs))
    for r, y1 in zip(recs, singles):
        assert torch.equal(r["y"], y1)
    assert not torch.equal(singles[0], singles[1])


CASES_F32 = [
    # Cin, Cout, H, W, ksize, stride, pad
    (64, 64, 24, 20, 3, 1, 1),
    (64, 3, 24, 20, 3, 1, 1),
    (64, 1, 17, 23, 3, 1, 1),
    (64, 2, 188, 188, 3, 1, 1),         # the head's last convs at full size: the plain-VALU kernel (<= 4 output channels)
    (128, 4, 35, 50, 3, 1, 1),
    (16, 3, 5, 7, 3, 1, 1),             # one channel chunk, an image smaller than a tile
    (128, 256, 20, 24, 3, 2, 1),
    (512, 64, 12, 16, 3, 1, 1),
    (128, 128, 33, 31, 1, 1, 0),
]


@pytest.mark.parametrize("Cin,Cout,H,W,k,stride,pad", CASES_F32)
def test_conv2d_f32(dev, Cin, Cout, H, W, k, stride, pad):
    from amdq3d import ops
    g_ = torch.Generator().manual_seed(Cin + Cout + H + k)
    x = torch.randn(1, Cin, H, W, generator=g_)
    w = torch.randn(Cout, Cin, k, k, generator=g_) / np.sqrt(k * k * Cin)
    b = torch.randn(Cout, generator=g_) * 0.1
    ps = torch.rand(Cout, generator=g_) + 0.5
    pt = torch.randn(Cout, generator=g_) * 0.1
    ref = torch.relu((F.conv2d(x, w, b, stride, pad)) * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))
    out_ld = max(4, (Cout + 3) // 4 * 4) + 4
    geom = ops.conv_geom(H, W, Cin, Cout, k, stride, pad, out_ld=out_ld, out_coff=4, relu=True)
    buf = torch.full((geom.Ho, geom.Wo, out_ld), -7.0, device=dev)
    ops.conv2d_f32(_nhwc(x).to(dev), ops.conv_weight_to_k(w).to(dev), geom, b.to(dev), ps.to(dev), pt.to(dev), out=buf)
    y = buf.cpu()
    assert float((y[..., :4] + 7.0).abs().max()) == 0.0          # channels outside [coff, coff+Cout) untouched
    y = y[..., 4:4 + Cout].permute(2, 0, 1)[None]
    assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("Cin,Cout,H,W,d", [(128, 256, 20, 24, 1), (256, 256, 10, 12, 2), (64, 32, 7, 9, 2)])
def test_deconv_f32(dev, Cin, Cout, H, W, d):
    from amdq3d import ops
    g_ = torch.Generator().manual_seed(Cin + Cout + d)
    x = torch.randn(1, Cin, H, W, generator=g_)
    w = torch.randn(Cin, Cout, d, d, generator=g_) / np.sqrt(Cin)
    ps = torch.rand(Cout, generator=g_) + 0.5
    pt = torch.randn(Cout, generator=g_) * 0.1
    ref = torch.relu(F.conv_transpose2d(x, w, None, stride=d) * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))
    geom = ops.conv_geom(H, W, Cin, Cout, 1, 1, 0, out_ld=2 * Cout, out_coff=Cout, deconv=d, relu=True)
    buf = torch.zeros((H * d, W * d, 2 * Cout), device=dev)
    po = torch.zeros((16, 2 * Cout), device=dev)
    ops.conv2d_f32(_nhwc(x).to(dev), ops.deconv_weight_to_k(w).to(dev), geom, None, ps.to(dev), pt.to(dev), out=buf, pieces_out=po)
    if H * d >= 3 and W * d >= 3:
        np.testing.assert_array_equal(po.cpu().numpy(), ops.piece_absmax(buf, H * d, W * d, 2 * Cout).cpu().numpy())
    y = buf.cpu()[..., Cout:].permute(2, 0, 1)[None]
    assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    if Cin % 32 == 0:
        # the same GEMM with the weights split into fp16 planes beforehand (what the engine hands the deblocks): the same products
        # in the same order — identical output and piece maxima
        wk = ops.deconv_weight_to_k(w).to(dev)
        buf2, po2 = torch.zeros_like(buf), torch.zeros_like(po)
        ops.conv2d_f32(_nhwc(x).to(dev), wk, geom, None, ps.to(dev), pt.to(dev), out=buf2, pieces_out=po2, weight_split=ops.split_weight(wk))
        np.testing.assert_array_equal(buf2.cpu().numpy(), buf.cpu().numpy())
        np.testing.assert_array_equal(po2.cpu().numpy(), po.cpu().numpy())


def _head_maps(H, W, ncls, seed, peaks=300):
    rng = np.random.default_rng(seed)
    hm = rng.normal(-4.0, 1.0, (ncls, H, W)).astype(np.float32)
    for _ in range(peaks):
        hm[rng.integers(0, ncls), rng.integers(0, H), rng.integers(0, W)] = rng.uniform(-1.0, 4.0)
    hm[0, 3, 5] = hm[ncls - 1, 7, 9] = hm[0, 11, 2] = 2.5            # exact score ties (across classes and inside one)
    center = rng.uniform(0, 1, (2, H, W)).astype(np.float32)
    center_z = rng.uniform(-1.5, 3.5, (1, H, W)).astype(np.float32)
    dim = rng.normal(0.5, 0.4, (3, H, W)).astype(np.float32)
    rot = rng.normal(0, 1, (2, H, W)).astype(np.float32)
    return hm, center, center_z, dim, rot


@pytest.mark.parametrize("H,W,ncls,K", [(47, 47, 3, 100), (188, 188, 3, 500), (20, 12, 1, 500)])
def test_center_decode(dev, H, W, ncls, K):
    from amdq3d import ops
    hm, center, center_z, dim, rot = _head_maps(H, W, ncls, seed=H + K)
    pc_range = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    b_ref, s_ref, l_ref = OD.decode_bbox_from_heatmap(OD.sigmoid(hm), center, center_z, np.exp(dim), rot, None, pc_range,
                                                      [0.1, 0.1, 0.15], 8, K, 0.1, pc_range)
    # device layout: one NHWC head buffer [H*W, 16]: hm 0.., center 4,5, z 6, dim 8..10, rot 12,13
    buf = np.zeros((H * W, 16), np.float32)
    buf[:, 0:ncls] = hm.reshape(ncls, -1).T
    buf[:, 4:6] = center.reshape(2, -1).T
    buf[:, 6:7] = center_z.reshape(1, -1).T
    buf[:, 8:11] = dim.reshape(3, -1).T
    buf[:, 12:14] = rot.reshape(2, -1).T
    t = torch.from_numpy(buf).to(dev)
    cfg = ops.make_decode_cfg(H, W, ncls, K, False, 8, [0.1, 0.1], pc_range, pc_range, 0.1, 16, 16)
    boxes, scores, labels, count = ops.center_decode(t, t[:, 4:], t[:, 6:], t[:, 8:], t[:, 12:], None, cfg)
    n = int(count.item())
    assert n == len(s_ref)
    np.testing.assert_array_equal(labels[:n].cpu().numpy(), l_ref)
    np.testing.assert_allclose(scores[:n].cpu().numpy(), s_ref, rtol=0, atol=1e-6)
    np.testing.assert_allclose(boxes[:n].cpu().numpy(), b_ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind,H,W,ncls,K", [("hover", 188, 188, 3, 500), ("constant", 188, 188, 3, 500), ("constant", 100, 100, 2, 500),
                                              ("steps", 100, 100, 3, 100), ("steps10", 100, 100, 2, 100), ("hover", 180, 180, 2, 83)])
@pytest.mark.parametrize("ws", ["full", "min"])
def test_center_decode_crowded_and_tied_maps(dev, kind, H, W, ncls, K, ws):
    """The top-K beyond the few-peaks case: more scores above the threshold than the candidate cache holds (the select then runs
    over every pixel's key), and maps full of equal scores (the K winners are the smallest pixel indices, across the chunks the
    score pass is split into) — random-weight heads produce the first, a saturated head the second."""
    from amdq3d import ops
    rng = np.random.default_rng(H + K + len(kind))
    hm, center, center_z, dim, rot = _head_maps(H, W, ncls, seed=H + K, peaks=0)
    if kind == "hover":
        hm = rng.normal(-2.19, 0.5, (ncls, H, W)).astype(np.float32)
    elif kind == "constant":
        hm = np.full((ncls, H, W), -1.0, np.float32)
    elif kind == "steps10":     # ten levels, ~1000 pixels each: the K-th score's histogram bin holds more keys than the rank-by-counting path takes
        hm = (rng.integers(0, 10, (ncls, H, W)) * 0.3 - 2.0).astype(np.float32)
    else:       # a handful of distinct levels: thousands of ties at the K-th score
        hm = rng.integers(-3, 2, (ncls, H, W)).astype(np.float32)
    pc_range = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    b_ref, s_ref, l_ref = OD.decode_bbox_from_heatmap(OD.sigmoid(hm), center, center_z, np.exp(dim), rot, None, pc_range,
                                                      [0.1, 0.1, 0.15], 8, K, 0.1, pc_range)
    buf = np.zeros((H * W, 24), np.float32)
    buf[:, 0:ncls] = hm.reshape(ncls, -1).T
    buf[:, 4:6] = center.reshape(2, -1).T
    buf[:, 6:7] = center_z.reshape(1, -1).T
    buf[:, 8:11] = dim.reshape(3, -1).T
    buf[:, 12:14] = rot.reshape(2, -1).T
    t = torch.from_numpy(buf).to(dev)
    cfg = ops.make_decode_cfg(H, W, ncls, K, False, 8, [0.1, 0.1], pc_range, pc_range, 0.1, 24, 24)
    # "min": the smallest workspace the call accepts — the one-workgroup-per-class top-K (what maps beyond 36 096 pixels get)
    boxes, scores, labels, count = ops.center_decode(t, t[:, 4:], t[:, 6:], t[:, 8:], t[:, 12:], None, cfg,
                                                     workspace="min" if ws == "min" else None)
    n = int(count.item())
    assert n == len(s_ref) and n > 0
    np.testing.assert_array_equal(labels[:n].cpu().numpy(), l_ref)
    np.testing.assert_allclose(scores[:n].cpu().numpy(), s_ref, rtol=0, atol=1e-6)
    np.testing.assert_allclose(boxes[:n].cpu().numpy(), b_ref, rtol=1e-5, atol=1e-5)


def _boxes(n, seed):
    rng = np.random.default_rng(seed)
    ctr = rng.uniform(-30, 30, (max(n // 6, 1), 2))
    b = np.zeros((n, 7), np.float32)
    which = rng.integers(0, len(ctr), n)
    b[:, 0:2] = ctr[which] + rng.normal(0, 0.6, (n, 2))
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3] = rng.uniform(3.5, 5.0, n); b[:, 4] = rng.uniform(1.6, 2.2, n); b[:, 5] = rng.uniform(1.4, 1.9, n)
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    return b


@pytest.mark.parametrize("n,thresh,post", [(500, 0.7, 500), (500, 0.2, 83), (65, 0.1, 500), (1, 0.7, 500), (1000, 0.5, 500)])
def test_nms_rotated_keep_list(dev, n, thresh, post):
    from amdq3d import ops
    b = _boxes(n, seed=n)
    ref = OD.nms_sorted(b, thresh)[:post]
    cap = n + 12
    t = torch.zeros((cap, 7), device=dev); t[:n] = torch.from_numpy(b).to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    keep, kc = ops.nms_rotated(t, cnt, thresh, post)
    m = int(kc.item())
    assert m == len(ref)
    np.testing.assert_array_equal(keep[:m].cpu().numpy(), ref)
    iou_dev_vs_ref = OD.iou_bev(b[:64], b[:64])
    assert np.isfinite(iou_dev_vs_ref).all()


def test_nms_exact_duplicates_are_suppressed(dev):
    """One pixel ranked in two classes yields the same BEV rectangle twice (the regression maps are shared). The polygon
    clipping of a rectangle with itself is degenerate (its area was decided by last-bit ties); the overlap is the area."""
    from amdq3d import ops
    n = 300
    b = _boxes(n, seed=7)
    b[5] = [-60.930511474609375, -43.97422790527344, 1.1041380167007446, 0.350118488073349, 1.073858618736267,
            2.750863552093506, 1.225360631942749]          # the pair that flipped on the device
    ref = OD.nms_sorted(b, 0.7)
    dup = np.concatenate([b, b[:40]], 0)                     # copies come later in rank order: none may survive
    t = torch.from_numpy(dup).to(dev)
    cnt = torch.tensor([len(dup)], dtype=torch.int32, device=dev)
    for _ in range(3):
        keep, kc = ops.nms_rotated(t, cnt, 0.7, 500)
        m = int(kc.item())
        np.testing.assert_array_equal(keep[:m].cpu().numpy(), ref)


def test_nms_empty_and_gather(dev):
    from amdq3d import ops
    t = torch.zeros((16, 7), device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    keep, kc = ops.nms_rotated(t, cnt, 0.7, 10)
    assert int(kc.item()) == 0
    b = _boxes(40, 3)
    t = torch.from_numpy(b).to(dev)
    scores = torch.linspace(0.9, 0.2, 40, device=dev)
    labels = (torch.arange(40, device=dev) % 3).to(torch.int32)
    cnt = torch.tensor([40], dtype=torch.int32, device=dev)
    keep, kc = ops.nms_rotated(t, cnt, 0.5, 500)
    cmap = torch.tensor([2, 0, 1], dtype=torch.int32, device=dev)
    bo, so, lo = ops.gather_dets(t, scores, labels, cmap, keep, kc)
    m = int(kc.item())
    k = keep[:m].long()
    assert torch.equal(bo[:m], t[k]) and torch.equal(so[:m], scores[k])
    assert torch.equal(lo[:m], cmap[labels[k].long()].long() + 1)


def test_decode_nms_batch_equals_head_by_head(dev):
    """aq3d_decode_nms_batch (five launches for all heads) against aq3d_center_decode + aq3d_nms_rotated + aq3d_gather_dets head by
    head: three heads with different class counts, map sizes, K and NMS capacities — every output bit for bit."""
    from amdq3d import ops, lib
    L = lib.load()
    pc_range = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    recs, singles = [], []
    for hi, (H, W, ncls, K, pre, post, thr) in enumerate([(47, 47, 3, 100, 80, 50, 0.2), (188, 188, 1, 500, 500, 83, 0.7), (20, 12, 2, 64, 64, 500, 0.1)]):
        hm, center, center_z, dim, rot = _head_maps(H, W, ncls, seed=H + K)
        buf = np.zeros((H * W, 16), np.float32)
        buf[:, 0:ncls] = hm.reshape(ncls, -1).T
        buf[:, 4:6] = center.reshape(2, -1).T
        buf[:, 6:7] = center_z.reshape(1, -1).T
        buf[:, 8:11] = dim.reshape(3, -1).T * 0.3 + 1.0           # boxes large enough to overlap
        buf[:, 12:14] = rot.reshape(2, -1).T
        t = torch.from_numpy(buf).to(dev)
        cfg = ops.make_decode_cfg(H, W, ncls, K, False, 8, [0.1, 0.1], pc_range, pc_range, 0.1, 16, 16)
        cmap = torch.arange(ncls, dtype=torch.int32, device=dev).flip(0).contiguous()
        cap = min(K, pre)
        # head by head
        bx, sc, lb, cnt = ops.center_decode(t, t[:, 4:], t[:, 6:], t[:, 8:], t[:, 12:], None, cfg)
        keep, kc = ops.nms_rotated(bx[:cap], cnt, thr, post)
        bo, so, lo = ops.gather_dets(bx[:cap], sc[:cap], lb[:cap], cmap, keep, kc)
        singles.append((bx, sc, lb, cnt, keep, kc, bo, so, lo))
        z = lambda *shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=dev)
        recs.append(dict(hm=t, center=t[:, 4:], center_z=t[:, 6:], dim=t[:, 8:], rot=t[:, 12:], vel=None, cfg=cfg,
                         cand=(z(K, 7), z(K), z(K, dt=torch.int32), z(1, dt=torch.int32)),
                         dec_ws=torch.empty((L.aq3d_center_decode_workspace_bytes(cfg),), dtype=torch.uint8, device=dev),
                         thresh=thr, nms_cap=cap, post_max=post, keep=z(cap, dt=torch.int32), keep_count=z(1, dt=torch.int32),
                         nms_ws=torch.empty((L.aq3d_nms_workspace_bytes(cap),), dtype=torch.uint8, device=dev), class_map=cmap,
                         out=(z(cap, 7), z(cap), z(cap, dt=torch.int64))))
    ops.decode_nms_batch(ops.head_problems(recs))
    for r, (bx, sc, lb, cnt, keep, kc, bo, so, lo) in zip(recs, singles):
        n, m = int(cnt.item()), int(kc.item())
        assert n > 10 and 0 < m <= n
        assert int(r["cand"][3].item()) == n and int(r["keep_count"].item()) == m
        assert torch.equal(r["cand"][0][:n], bx[:n]) and torch.equal(r["cand"][1][:n], sc[:n]) and torch.equal(r["cand"][2][:n], lb[:n])
        assert torch.equal(r["keep"][:m], keep[:m])
        assert torch.equal(r["out"][0][:m], bo[:m]) and torch.equal(r["out"][1][:m], so[:m]) and torch.equal(r["out"][2][:m], lo[:m])


def test_boxes_iou3d_and_recall_record(dev):
    """boxes_iou3d_gpu + generate_recall_record (detector3d_template.py:286-328) against the oracle."""
    from amdq3d import ops
    from amdq3d.detector import CenterPoint
    a, b = _boxes(120, 7), _boxes(40, 8)
    b[:15] = a[:15] + np.random.default_rng(0).normal(0, 0.05, (15, 7)).astype(np.float32)     # some well-matched pairs
    ref = OD.boxes_iou3d(a, b)
    got = ops.boxes_iou3d(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    assert ops.boxes_iou3d(torch.zeros((0, 7), device=dev), torch.from_numpy(b).to(dev)).shape == (0, 40)
    gt = np.zeros((2, 50, 8), np.float32)
    gt[0, :40, :7] = b; gt[0, :40, 7] = 1
    gt[1, :10, :7] = b[:10]; gt[1, :10, 7] = 2
    data = {"gt_boxes": torch.from_numpy(gt).to(dev), "batch_size": 2}
    rd = {}
    preds = [torch.from_numpy(a).to(dev), torch.from_numpy(a[:30]).to(dev)]
    for bi in range(2):
        rd = CenterPoint.generate_recall_record(preds[bi], rd, bi, data, [0.3, 0.5, 0.7])
    exp = {"gt": 50}
    for t in (0.3, 0.5, 0.7):
        exp["rcnn_%s" % t] = int((ref.max(axis=0) > t).sum()) + int((OD.boxes_iou3d(a[:30], b[:10]).max(axis=0) > t).sum())
        exp["roi_%s" % t] = 0
    assert rd == exp
    assert CenterPoint.generate_recall_record(preds[0], {}, 0, {"batch_size": 1}, [0.3]) == {}
