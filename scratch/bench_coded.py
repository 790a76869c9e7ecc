"""Per-layer timing on the real W150k index levels: round-1 path (aq3d_quantize_rows(split) + aq3d_spconv f16-split kernels)
vs the coded path (aq3d_spconv_prep + aq3d_spconv_coded). Also checks that both give the same values."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d
import numpy as np
import torch, bench
from amdq3d import ops
from amdq3d.datasets import load_data_to_gpu

dev = torch.device("cuda:0")
os.environ["AQ3D_GRAPH"] = "0"
cfgp, npts, nvox, _ = bench.WORKLOADS[os.environ.get("WORKLOAD", "centerpoint")]
cfg, ds, model, state = bench.build(cfgp, dev, npts, nvox)
b = ds.collate_batch([ds[0]]); load_data_to_gpu(b, dev)
with torch.no_grad():
    model(b)
eng = model.engine()
torch.cuda.synchronize()
ITERS = 30


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / ITERS


tot_old = tot_new = tot_tiled = 0.0
for rec in eng.sparse_convs:
    q = rec["quant"]
    if q is None or not q["on"] or rec.get("weight_i8") is None:
        continue
    cin, cout, kvol = rec["cin"], rec["cout"], rec["kvol"]
    if not ops.spconv_coded_supported(cin, cout, kvol):
        print(rec["path"], "unsupported by the coded path"); continue
    x = rec["x"]
    lv_in, lv_out = eng.levels[x["level"]], eng.levels[rec["level"]]
    nbr = eng._nbr_of(rec["nbr"], rec["level"])
    cnt_in, cnt_out = lv_in["count"], lv_out["count"]
    amax = torch.zeros((cin,), device=dev)
    ops.absmax_cols(x["buf"], cnt_in, amax)
    res = None if rec["residual"] is None else rec["residual"]["buf"]
    out_a = torch.zeros_like(rec["out"]["buf"]); out_b = torch.zeros_like(rec["out"]["buf"])
    split = torch.zeros((eng.cap, 2 * cin), dtype=torch.float16, device=dev)
    dq = torch.zeros((cin,), device=dev)
    L = ops._lib.load()

    def q_old():
        ops.check(L.aq3d_quantize_rows(ops._p(x["buf"]), ops._p(amax), ops._p(cnt_in), eng.cap, cin, 8, 0, None, None, ops._p(dq),
                                       ops._p(split), ops._stream()), "q")

    def c_old():
        ops.spconv(split, rec["weight"], nbr, cnt_out, cout, w_dequant=rec["w_dequant"], bias=rec["bias"], post_scale=rec["post_scale"],
                   post_shift=rec["post_shift"], residual=res, relu=rec["relu"], out=out_a)

    bufs = (torch.zeros((eng.cap, cin), dtype=torch.int8, device=dev), torch.empty((kvol, cout, 2, cin), dtype=torch.float16, device=dev),
            torch.empty((cout,), device=dev))

    def q_new():
        ops.spconv_prep(x["buf"], amax, cnt_in, rec["weight_i8"], rec["w_dequant"], out=bufs)

    def c_new():
        ops.spconv_coded(bufs[0], bufs[1], bufs[2], nbr, cnt_out, bias=rec["bias"], post_scale=rec["post_scale"],
                         post_shift=rec["post_shift"], residual=res, relu=rec["relu"], out=out_b)

    perm, nbr_t = ops.tile_order(nbr, cnt_out)
    out_c = torch.zeros_like(rec["out"]["buf"])

    def c_tiled():
        ops.spconv_coded(bufs[0], bufs[1], bufs[2], nbr_t, cnt_out, perm=perm, bias=rec["bias"], post_scale=rec["post_scale"],
                         post_shift=rec["post_shift"], residual=res, relu=rec["relu"], out=out_c)

    def t_order():
        ops.tile_order(nbr, cnt_out, out=(perm, nbr_t))

    q_old(); c_old(); q_new(); c_new(); c_tiled()
    torch.cuda.synchronize()
    assert torch.equal(out_b[:int(cnt_out.item())], out_c[:int(cnt_out.item())]), "tile order changed values"
    m = int(cnt_out.item())
    d = (out_a[:m] - out_b[:m]).abs().max().item() / max(1.0, out_a[:m].abs().max().item())
    t = [timeit(f) for f in (q_old, c_old, q_new, c_new, c_tiled, t_order)]
    pairs = int((nbr[:m] >= 0).sum().item())
    tot_old += t[0] + t[1]; tot_new += t[2] + t[3]; tot_tiled += t[2] + t[4]
    print("%-30s %3d->%3d k%-2d rows %6d pairs/row %4.1f | old quant %5.1f conv %6.1f | new prep %5.1f conv %6.1f tiled %6.1f (sort %5.1f) us | rel diff %.1e"
          % (rec["path"][12:], cin, cout, kvol, m, pairs / max(m, 1), t[0], t[1], t[2], t[3], t[4], t[5], d), flush=True)
print("sum over quantised layers: old %.1f us, new %.1f us, new in tile order %.1f us" % (tot_old, tot_new, tot_tiled))
