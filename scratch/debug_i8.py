import sys, numpy as np, torch
sys.path.insert(0, '.')
from amdq3d import ops
from oracle import quantizer as Q
import torch.nn.functional as F
dev = torch.device('cuda:0')
for (Cin, Cout, H, W, stride, alpha) in [(128, 256, 36, 40, 2, 0.5), (128, 128, 40, 36, 1, 0.5), (128,128,36,40,2,0.5)]:
    g_ = torch.Generator().manual_seed(Cin * 3 + Cout + H)
    x = torch.relu(torch.randn(1, Cin, H, W, generator=g_)) * (torch.rand(1, Cin, 1, 1, generator=g_) * 3 + 0.1)
    x[:, :, 0, :] *= 2.5
    x[:, :, :, -1] *= 1.7
    w = torch.randn(Cout, Cin, 3, 3, generator=g_) / np.sqrt(9 * Cin)
    # oracle, with K order (kh,kw,c) to compare with device params
    cols = F.unfold(x, 3, padding=1, stride=stride).permute(0, 2, 1).reshape(-1, Cin * 9)   # (c,kh,kw)
    w2 = w.reshape(Cout, -1)
    w_scale = w2.abs().max(0)[0]; a_scale = cols.abs().max(0)[0]
    scale = a_scale ** alpha / w_scale ** (1 - alpha); scale[scale == 0] = 1
    perm = torch.arange(Cin * 9).view(Cin, 9).t().reshape(-1)     # device col j=(tap,c) → ref col c*9+tap
    ws = (w2 * scale); xs_ = cols / scale
    aw = Q.reduce_amax(ws.numpy(), axis=0); ax = Q.reduce_amax(xs_.numpy())
    _, wcodes, wdiv = Q.fake_quant(ws.numpy(), aw, 8, return_codes=True)
    _, xcodes, xdiv = Q.fake_quant(xs_.numpy(), ax, 8, return_codes=True)
    xg = x[0].permute(1, 2, 0).contiguous().to(dev)
    geom = ops.conv_geom(H, W, Cin, Cout, 3, stride, 1)
    wk = ops.conv_weight_to_k(w).to(dev); wcol = wk.abs().max(0)[0].contiguous()
    pieces = ops.piece_absmax(xg, H, W, Cin)
    params = ops.sq_prep(pieces, wk, wcol, geom, alpha).cpu()
    K = 9 * Cin; al = lambda v: (v + 255) // 256 * 256
    s_dev = params[:K * 4].view(torch.float32)
    mul = params[2 * al(K * 4): 2 * al(K * 4) + Cout * 4].view(torch.float32)
    scal = params[2 * al(K * 4) + al(Cout * 4):][:32].view(torch.float32)
    wq = params[2 * al(K * 4) + al(Cout * 4) + 256:][:Cout * K].view(torch.int8).view(Cout, K)
    s_ref = scale[perm]
    print('case', Cin, Cout, H, W, stride)
    print(' s mismatch bits:', int((s_dev.view(torch.int32) != s_ref.view(torch.int32)).sum()), 'max rel', float(((s_dev - s_ref).abs() / s_ref).max()))
    a_dev = None
    print(' amax_x dev', scal[1].item(), 'ref', float(ax), 'sx dev', scal[0].item())
    wc_ref = torch.from_numpy(wcodes)[:, perm]
    print(' weight code mismatches:', int((wq.float() != wc_ref).sum()), 'of', wq.numel())
    y = ops.conv2d_i8(xg, params.to(dev), geom).cpu()
    yi_ref = torch.from_numpy(xcodes) @ torch.from_numpy(wcodes).T      # exact ints in fp32? use float64
    yi_ref = (torch.from_numpy(xcodes).double() @ torch.from_numpy(wcodes).double().T)
    y_int_dev = (y.view(-1, Cout).double() / mul.double()[None, :])
    d = (y_int_dev - yi_ref).abs()
    print(' int-domain max diff', float(d.max()), 'n>0.5:', int((d > 0.5).sum()), 'of', d.numel())
    bad = (d > 0.5).nonzero()[:10]
    print(' bad (pix,oc):', bad.tolist())
