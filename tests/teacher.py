"""Teacher-forced per-layer parity (test infrastructure): every conv of a compiled plan is fed the ORACLE's input tensors
and its output is compared with the oracle's output of the same layer. A dynamically quantised network is chaotic in the
last bit end to end; layer by layer on identical inputs it is not, so this is where the int8 kernels are
held to a tight bar at the sizes the benchmark runs.

The engine must have run the frame once (index levels built). Rows of a sparse tensor are matched by linear site key:
the engine keeps first-seen / emission order, the oracle keeps key order.
"""
import numpy as np
import torch

from oracle import sparse as OS


def row_maps(eng, oracle_index_sets):
    """oracle_index_sets[li] = i32 [M, 4] rows (b, z, y, x) of engine level li → list of maps: engine row r is oracle row map[r]."""
    maps = []
    for li, idx in enumerate(oracle_index_sets):
        lv = eng.levels[li]
        m = int(lv["count"].item())
        assert m == len(idx), "level %d: %d live rows on the device, %d in the oracle" % (li, m, len(idx))
        ek = OS._key(lv["coords"][:m].cpu().numpy(), lv["shape"])
        ok = OS._key(np.asarray(idx), lv["shape"])
        order = np.argsort(ok, kind="stable")
        pos = np.searchsorted(ok[order], ek)
        pos = np.minimum(pos, len(ok) - 1)
        assert np.array_equal(ok[order][pos], ek), "level %d: site sets differ" % li      # bit-exact active set
        maps.append(order[pos])
    return maps


def _rel(out, ref):
    """max |out - ref| in units of the layer's output scale (max |ref|, at least 1)."""
    return float(np.abs(out.astype(np.float64) - ref.astype(np.float64)).max() / max(1.0, float(np.abs(ref).max())))


def sparse_layers(eng, maps, oracle_rows, steps=None):
    """Runs every sparse conv (single and grouped steps) on the oracle's inputs → [(path, kind, rel_err, frac_gt_1e-4)].
    oracle_rows(name) → f32 [M, C] rows of the named tensor in ORACLE order ('vfe', conv paths, 'backbone_3d.bev_out')."""
    from amdq3d import ops
    dev = eng.device
    res = []
    eng._force_rows = True        # static plans: the forced input has to be quantised here, not by a producer epilogue

    def put(t):
        rows = oracle_rows(t["name"])
        mp = maps[t["level"]]
        assert rows.shape[0] == len(mp), (t["name"], rows.shape, len(mp))
        buf = t["buf"]
        buf[:len(mp)].zero_()
        buf[:len(mp), :rows.shape[1]] = torch.from_numpy(np.ascontiguousarray(rows[mp])).to(dev)

    def prepare(rec):
        put(rec["x"])
        if rec["residual"] is not None:
            put(rec["residual"])
        q = rec["quant"]
        if q is not None and (q["on"] or q["calib"]):
            x = rec["x"]
            ops.absmax_cols(x["buf"], eng.levels[x["level"]]["count"], eng.arena.view(x["amax"]))

    def check(rec):
        mp = maps[rec["level"]]
        ref = oracle_rows(rec["path"])[mp]
        out = rec["out"]["buf"][:len(mp), :ref.shape[1]].cpu().numpy()
        kind = "f32" if rec["w_dequant"] is None else ("w8a8" if rec["quant"] is not None and rec["quant"]["on"] else "w8")
        d = np.abs(out - ref) / max(1.0, float(np.abs(ref).max()))
        res.append((rec["path"], "%s %d->%d k%d rows %d" % (kind, rec["cin"], rec["cout_real"], rec["kvol"], len(mp)),
                    float(d.max()) if d.size else 0.0, float((d > 1e-4).mean()) if d.size else 0.0))

    for kind, arg in (steps if steps is not None else eng.steps):
        if kind == "spconv":
            eng.arena.clear()
            prepare(arg)
            eng._run_spconv(arg)
            check(arg)
        elif kind == "spconv_group":
            eng.arena.clear()
            for r in arg["recs"]:
                prepare(r)
            eng._run_spconv_group(arg)
            for r in arg["recs"]:
                check(r)
    torch.cuda.synchronize()
    eng._force_rows = False
    return res


def dense_layers(eng, oracle_nchw, B=1):
    """Runs every dense conv (single and grouped steps) on the oracle's inputs → [(path, kind, rel_err, frac_gt_1e-4)].
    oracle_nchw(name) → torch f32 [B, C, H, W]: 'spatial_features', 'spatial_features_2d' or a conv path (post BN/ReLU)."""
    from amdq3d import ops
    dev = eng.device
    producers = {}
    for r in eng.dense_convs:
        producers.setdefault(r["out"].data_ptr(), []).append(r)
    res = []

    def source_of(x):
        if x.data_ptr() == eng.spatial.data_ptr():
            return "spatial_features"
        if x.data_ptr() == eng.bev_cat.data_ptr() and len(producers.get(x.data_ptr(), [])) != 1:
            return "spatial_features_2d"
        prods = producers[x.data_ptr()]
        assert len(prods) == 1, [p["path"] for p in prods]
        return prods[0]["path"]

    def prepare(rec):
        x = rec["x"]
        t = oracle_nchw(source_of(x))                                   # [B, C, H, W]
        nhwc = t.permute(0, 2, 3, 1).contiguous()
        assert tuple(nhwc.shape[:3]) == tuple(x.shape[:3]) and nhwc.shape[3] <= x.shape[3], (rec["path"], nhwc.shape, x.shape)
        x[:B, :, :, :nhwc.shape[3]].copy_(nhwc.to(dev))
        if rec["kind"] in ("i8", "sq16"):
            g = rec["geom"]
            pieces = eng.arena.view(rec["pieces"], (16, g.Cin))
            for b in range(B):
                ops.piece_absmax(x[b], g.H, g.W, g.Cin, g.in_ld, pieces)

    def check(rec):
        ref = oracle_nchw(rec["path"]).permute(0, 2, 3, 1).numpy()     # [B, Ho, Wo, Cout]
        g = rec["geom"]
        out = rec["out"][:B, :, :, g.out_coff:g.out_coff + ref.shape[3]].cpu().numpy()
        assert out.shape == ref.shape, (rec["path"], out.shape, ref.shape)
        d = np.abs(out - ref) / max(1.0, float(np.abs(ref).max()))
        res.append((rec["path"], "%s %d->%d k%d s%d %dx%d" % (rec["kind"], g.Cin, ref.shape[3], g.ksize, g.stride, g.H, g.W),
                    float(d.max()), float((d > 1e-4).mean())))

    for kind, arg in eng.steps:
        if kind == "conv2d":
            eng.arena.clear()
            prepare(arg)
            eng._run_conv2d(arg, B)
            check(arg)
        elif kind == "conv2d_group":
            eng.arena.clear()
            for r in arg["recs"]:
                prepare(r)
            eng._run_conv2d_group(arg, B)
            for r in arg["recs"]:
                check(r)
    torch.cuda.synchronize()
    return res


def report(title, res):
    print("\n%s" % title)
    for path, kind, err, frac in res:
        print("  %-52s %-34s max %.3e  frac>1e-4 %.2e" % (path, kind, err, frac))
