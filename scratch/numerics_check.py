"""Where does a build differ numerically? Stage-by-stage maxima against float64 numpy for the quantiser, the W' builder and
the sparse conv kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import amdq3d
import numpy as np, torch
from amdq3d import ops, lib
if os.environ.get('AQ3D_LIB'):
    lib.load(os.environ['AQ3D_LIB']); print('library:', os.environ['AQ3D_LIB'])
from oracle import sparse as OS
dev = torch.device("cuda:0")
for cin, cout in ((16, 16), (64, 64), (128, 128)):
    rng = np.random.default_rng(cin)
    shape = [41, 160, 160]
    idx = np.unique(np.stack([np.zeros(2500, np.int64), rng.integers(0, 12, 2500), rng.integers(0, 60, 2500), rng.integers(0, 60, 2500)], 1), axis=0).astype(np.int32)
    m = len(idx)
    feats = np.maximum(rng.normal(0, 1, (m, cin)), 0).astype(np.float32) * rng.uniform(0.1, 4, cin).astype(np.float32)
    w = (rng.normal(0, 1, (cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    wq, wcodes, wdiv = OS.qconv_weight(w, 8)
    aq, amax_ref = OS.qconv_act(feats, 8, cw=True)
    nbr_ref = OS.subm_neighbors(idx, shape)
    ref = OS.gather_conv(aq, wq, nbr_ref)
    ref64 = np.zeros((m, cout))
    w64 = wq.reshape(cout, 27, cin).astype(np.float64)
    for k in range(27):
        rows = np.nonzero(nbr_ref[:, k] >= 0)[0]
        ref64[rows] += aq[nbr_ref[rows, k]].astype(np.float64) @ w64[:, k, :].T
    cap = m + 5
    nbr = torch.full((cap, 27), -1, dtype=torch.int32, device=dev); nbr[:m] = torch.from_numpy(nbr_ref).to(dev)
    count = torch.tensor([m], dtype=torch.int32, device=dev)
    fin = torch.zeros((cap, cin), device=dev); fin[:m] = torch.from_numpy(feats).to(dev)
    amax = ops.absmax_cols(fin, count)
    codes, fq, dq = ops.quantize_rows(fin, amax, count, bits=8, want_codes=True, want_fq=True)
    fqh = fq[:m].cpu().numpy()
    print("C=%d  fq vs oracle: %.2e" % (cin, np.abs(fqh - aq).max()))
    wk8 = torch.from_numpy(np.ascontiguousarray(wcodes.reshape(cout, 27, cin).transpose(1, 0, 2)).astype(np.int8)).to(dev)
    wdq = torch.from_numpy((np.float32(1.0) / wdiv).astype(np.float32)).to(dev)
    cds, wprep, out_mul = ops.spconv_prep(fin, amax, count, wk8, wdq)
    dqh = dq.cpu().numpy().astype(np.float64)
    S = dqh.max()
    wp = wprep.float().cpu().numpy().astype(np.float64)            # [27, cout, 2, cin]
    exact = wk8.cpu().numpy().astype(np.float64) * (dqh / S)[None, None, :]
    print("      W' hi+lo vs exact: %.2e (|W'| max %.1f); hi alone %.2e; out_mul rel err %.2e" % (
        np.abs(wp[:, :, 0] + wp[:, :, 1] - exact).max(), np.abs(exact).max(), np.abs(wp[:, :, 0] - exact).max(),
        np.abs(out_mul.cpu().numpy() - S * wdq.cpu().numpy().astype(np.float64)).max() / (S * wdq.cpu().numpy().max())))
    scale = max(1.0, np.abs(ref64).max())
    y_coded = ops.spconv_coded(cds, wprep, out_mul, nbr, count)[:m].cpu().numpy()
    y_i8 = ops.spconv(codes, wk8, nbr, count, cout, act_dequant=dq, w_dequant=wdq)[:m].cpu().numpy()
    y_f32 = ops.spconv(fq, torch.from_numpy(np.ascontiguousarray(wq.reshape(cout, 27, cin).transpose(1, 0, 2))).to(dev), nbr, count, cout)[:m].cpu().numpy()
    for name, y in (("coded", y_coded), ("i8 generic", y_i8), ("f32", y_f32), ("oracle f32", ref)):
        d = np.abs(y - ref64) / scale
        print("      %-10s vs float64: max %.2e  mean %.2e  frac>1e-5 %.2e" % (name, d.max(), d.mean(), (d > 1e-5).mean()))
