import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch, numpy as np
    import bench
    from amdq3d.datasets import load_data_to_gpu
    dev = torch.device("cuda:0")
    cfg, ds, model, state = bench.build("cfgs/waymo_models/centerpoint.yaml", dev)
    out = []
    for i in range(8):
        b = ds.collate_batch([ds[i]]); load_data_to_gpu(b, dev)
        with torch.no_grad():
            pred, _ = model(b)
        p = pred[0]
        out.append((p["pred_boxes"].cpu().numpy(), p["pred_scores"].cpu().numpy(), p["pred_labels"].cpu().numpy()))
    np.save(sys.argv[1], np.array([np.concatenate([a.ravel(), b_.ravel(), c.ravel().astype(np.float32)]) for a, b_, c in out], dtype=object), allow_pickle=True)
    print([len(o[1]) for o in out])
else:
    import numpy as np
    for name, env in (("gpurun_out/t_f.npy", {}), ("gpurun_out/t_n.npy", {"AQ3D_TOPK_NOFILTER": "1"})):
        e = dict(os.environ); e.update(env); e["AQ3D_GRAPH"] = "0"
        print(name, subprocess.run([sys.executable, __file__, name], env=e, capture_output=True, text=True).stdout.strip())
    a = np.load("gpurun_out/t_f.npy", allow_pickle=True); b = np.load("gpurun_out/t_n.npy", allow_pickle=True)
    for i in range(8):
        same = a[i].shape == b[i].shape and np.array_equal(a[i], b[i])
        print("frame", i, "identical" if same else "DIFFERENT %s %s" % (a[i].shape, b[i].shape))
