import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import build_model, small_cfg
from oracle import centerpoint as OC, sparse as OS
from amdq3d import quant as Qz
from amdq3d.datasets import load_data_to_gpu
dev = torch.device('cuda:0')
for recipe in ['fp32', 'sq']:
    cfg = small_cfg('waymo')
    ds, model = build_model(cfg, 9000, 2500)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    orec = None
    if recipe == 'sq':
        Qz.quant(model, 8, 8, sq=True, alpha=0.5, static=False); orec = OC.Recipe(8, 8, True, 0.5)
    model.cuda()
    batch = ds.collate_batch([ds[0]]); load_data_to_gpu(batch, dev)
    with torch.no_grad(): pred, _ = model(batch)
    tr = {}
    OC.run(state, cfg, ds[0]['points'], orec, tr)
    eng = model.engine()
    print('=== recipe', recipe)
    lv_shapes = [lv['shape'] for lv in eng.levels]
    # oracle coords per level
    oidx = {0: None}
    levels = tr['levels']
    o_level_idx = [levels['x_conv1'][1], levels['x_conv2'][1], levels['x_conv3'][1], levels['x_conv4'][1], tr['encoded'][1]]
    for rec in eng.sparse_convs:
        ref = tr.get(rec['path'])
        if ref is None: print('no ref', rec['path']); continue
        li = rec['level']
        m = int(eng.levels[li]['count'].item())
        co = eng.levels[li]['coords'][:m].cpu().numpy()
        key_d = OS._key(co, lv_shapes[li]); key_o = OS._key(o_level_idx[li], lv_shapes[li])
        od = np.argsort(key_d); oo = np.argsort(key_o)
        assert np.array_equal(key_d[od], key_o[oo]), 'index set differs at %s' % rec['path']
        y = rec['out']['buf'][:m].cpu().numpy()[od]; r = ref[oo]
        print('%-34s max|d| %.3e  scale %.3g  frac>1e-4: %.4f' % (rec['path'], np.abs(y - r).max(), np.abs(r).max(), (np.abs(y - r) > 1e-4 * max(1, np.abs(r).max())).mean()))
    for rec in eng.dense_convs:
        ref = tr.get(rec['path'])
        if ref is None: continue
        y = rec['out'][0].cpu().numpy()[..., :ref.shape[1]]
        r = ref[0].permute(1, 2, 0).numpy()
        print('%-34s max|d| %.3e  scale %.3g  frac>1e-4: %.4f' % (rec['path'], np.abs(y - r).max(), np.abs(r).max(), (np.abs(y - r) > 1e-4 * max(1, np.abs(r).max())).mean()))
