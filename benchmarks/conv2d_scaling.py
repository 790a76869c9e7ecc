"""Dense conv kernels at the benchmark's BEV size: time against the number of K steps (fixed cost vs per-step cost).

    python benchmarks/conv2d_scaling.py            # one JSON line per case

conv2d_i8 (SQConv2d): 188x188 / 94x94 images, Cin in {64..512}, ksize in {1, 3} -> nk = ksize^2 * Cin / 64 steps per workgroup.
conv2d_f32 (fp16-split pipe): the two deblock shapes and the head's 3x3 fp32 convs.
HIP events on the launch stream, >= 30 ms of warm-up (clock ramp), median of 20 groups of 10 launches.
"""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdq3d  # noqa: F401,E402
import torch  # noqa: E402
from amdq3d import ops  # noqa: E402


def timed(fn, groups=20, per=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    n = 0
    while True:                                  # warm up for >= 30 ms
        for _ in range(per):
            fn()
        n += per
        t1.record()
        t1.synchronize()
        if t0.elapsed_time(t1) > 30.0:
            break
    ts = []
    for _ in range(groups):
        t0.record()
        for _ in range(per):
            fn()
        t1.record()
        t1.synchronize()
        ts.append(t0.elapsed_time(t1) * 1e3 / per)
    ts.sort()
    return ts[len(ts) // 2], ts[1], ts[-2]


def main():
    dev = torch.device("cuda:0")
    g_ = torch.Generator().manual_seed(3)
    for (H, W, Cin, Cout, k, s, pieces) in [
            (188, 188, 64, 128, 1, 1, True), (188, 188, 64, 128, 3, 1, True), (188, 188, 128, 128, 3, 1, True),
            (188, 188, 128, 128, 3, 1, False), (188, 188, 256, 128, 3, 1, True), (188, 188, 512, 64, 3, 1, False),
            (188, 188, 128, 256, 3, 2, True), (94, 94, 64, 256, 1, 1, True), (94, 94, 256, 256, 3, 1, True), (94, 94, 256, 256, 3, 1, False)]:
        x = (torch.relu(torch.randn(H, W, Cin, generator=g_)) * (torch.rand(Cin, generator=g_) * 3 + 0.1)).to(dev)
        w = torch.randn(Cout, Cin, k, k, generator=g_) / (k * k * Cin) ** 0.5
        geom = ops.conv_geom(H, W, Cin, Cout, k, s, k // 2, relu=True)
        wk = ops.conv_weight_to_k(w).to(dev)
        wcol = wk.abs().max(dim=0)[0].contiguous()
        params = ops.sq_prep(ops.piece_absmax(x, H, W, Cin), wk, wcol, geom, 0.5)
        ps, pt = (torch.rand(Cout, generator=g_) + 0.5).to(dev), (torch.randn(Cout, generator=g_) * 0.1).to(dev)
        y = torch.empty((geom.Ho, geom.Wo, Cout), device=dev)
        po = torch.zeros((16, Cout), device=dev) if pieces else None
        med, lo, hi = timed(lambda: ops.conv2d_i8(x, params, geom, post_scale=ps, post_shift=pt, out=y, pieces_out=po, pieces_mode=1 if pieces else 0))
        nk = k * k * Cin // 64
        print(json.dumps(dict(kernel="conv2d_i8", H=H, W=W, Cin=Cin, Cout=Cout, ksize=k, stride=s, pieces=pieces, k_steps=nk,
                              us=round(med, 2), us_p10=round(lo, 2), us_p90=round(hi, 2),
                              GOP=round(2e-9 * geom.Ho * geom.Wo * k * k * Cin * Cout, 2))), flush=True)
    for (H, W, Cin, Cout, k, d, pieces) in [(188, 188, 128, 256, 1, 1, True), (188, 188, 128, 256, 1, 1, False), (94, 94, 256, 256, 1, 2, True),
                                            (94, 94, 256, 256, 1, 2, False), (188, 188, 64, 64, 3, 0, False), (188, 188, 64, 64, 1, 0, False)]:
        x = torch.randn(H, W, Cin, generator=g_).to(dev)
        if d:
            wk = (torch.randn(d * d * Cout, Cin, generator=g_) / Cin ** 0.5).to(dev)
            geom = ops.conv_geom(H, W, Cin, Cout, 1, 1, 0, out_ld=512, out_coff=0, deconv=d, relu=True)
            y = torch.empty((H * d, W * d, 512), device=dev)
        else:
            wk = ops.conv_weight_to_k(torch.randn(Cout, Cin, k, k, generator=g_) / (k * k * Cin) ** 0.5).to(dev)
            geom = ops.conv_geom(H, W, Cin, Cout, k, 1, k // 2, relu=True)
            y = torch.empty((H, W, Cout), device=dev)
        ps, pt = (torch.rand(Cout, generator=g_) + 0.5).to(dev), (torch.randn(Cout, generator=g_) * 0.1).to(dev)
        po = torch.zeros((16, 512), device=dev) if pieces else None
        med, lo, hi = timed(lambda: ops.conv2d_f32(x, wk, geom, None, ps, pt, out=y, pieces_out=po))
        print(json.dumps(dict(kernel="conv2d_f32", H=H, W=W, Cin=Cin, Cout=Cout, ksize=k, deconv=d, pieces=pieces, k_steps=k * k * Cin // 32,
                              us=round(med, 2), us_p10=round(lo, 2), us_p90=round(hi, 2))), flush=True)


if __name__ == "__main__":
    main()
