#!/usr/bin/env python
"""Counterpart of the reference's evaluation loop for the MI355X-native detector.

    tools/test.py:138-204 (argument handling, build_network, load_params_from_file, model.cuda()) and
    tools/eval_utils/eval_utils.py:20-139 (eval_one_epoch: load_data_to_gpu -> model(batch_dict) under no_grad ->
    statistics_info -> dataset.generate_prediction_dicts -> result.pkl, `--infer_time` meter with a 10 % warm-up).

What is kept: the record format of result.pkl (list of {'name', 'score', 'boxes_lidar', 'pred_labels', 'frame_id'}), the
recall bookkeeping keys ('recall/roi_<t>', 'recall/rcnn_<t>' from ret_dict['gt'|'roi_<t>'|'rcnn_<t>']), the per-rank frame
striding and the rank-0 merge. What is different by design: `--infer_time` brackets the forward with HIP events on the launch
stream (the reference takes wall time under CUDA_LAUNCH_BLOCKING, test.py:141-142); official dataset evaluators are out of
scope (no datasets in the image), so `dataset.evaluation` is replaced by the average number of predicted objects.

    python tools/eval.py --cfg_file cfgs/waymo_models/centerpoint.yaml [--ckpt model.pth] [--frames DIR | --synthetic N]
                         [--quant sq|plain|none] [--static N_CALIB] [--infer_time] [--result_dir out/]
    python -m torch.distributed.run --nproc-per-node 8 ... tools/eval.py --launcher pytorch ...
"""
import argparse
import logging
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import amdq3d  # noqa: E402,F401  (before torch touches the GPU: graph-replay flags)
import numpy as np  # noqa: E402
import torch  # noqa: E402


class AverageMeter:
    """common_utils.AverageMeter (pcdet/utils/common_utils.py:252-270)."""

    def __init__(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def statistics_info(cfg, ret_dict, metric, disp_dict):
    """eval_utils.py:11-17."""
    for cur_thresh in cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST:
        metric["recall_roi_%s" % str(cur_thresh)] += ret_dict.get("roi_%s" % str(cur_thresh), 0)
        metric["recall_rcnn_%s" % str(cur_thresh)] += ret_dict.get("rcnn_%s" % str(cur_thresh), 0)
    metric["gt_num"] += ret_dict.get("gt", 0)
    min_thresh = cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST[0]
    disp_dict["recall_%s" % str(min_thresh)] = "(%d, %d) / %d" % (metric["recall_roi_%s" % str(min_thresh)],
                                                                 metric["recall_rcnn_%s" % str(min_thresh)], metric["gt_num"])


def eval_one_epoch(cfg, args, model, dataset, indices, logger, result_dir, device, rank=0, world=1):
    """One pass over `indices` of `dataset`; returns (ret_dict, det_annos). Mirrors eval_utils.eval_one_epoch."""
    from amdq3d.datasets import load_data_to_gpu
    os.makedirs(result_dir, exist_ok=True)
    metric = {"gt_num": 0}
    for t in cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST:
        metric["recall_roi_%s" % str(t)] = 0
        metric["recall_rcnn_%s" % str(t)] = 0
    class_names = dataset.class_names
    det_annos = []
    meter = AverageMeter()
    start_iter = int(len(indices) * 0.1 / max(1, int(getattr(args, "batch_size", 1) or 1))) if args.infer_time else 0
    model.eval()
    t_epoch = time.time()
    bs = max(1, int(getattr(args, "batch_size", 1) or 1))        # tools/test.py --batch_size: frames per forward (the model's dynamic
    if bs > model.max_batch_size:                                 # quantiser statistics are then taken over the batch, as the reference's are)
        model.max_batch_size = bs
    groups = [list(indices[i:i + bs]) for i in range(0, len(indices), bs)]
    for it, grp in enumerate(groups):
        batch_dict = dataset.collate_batch([dataset[idx] for idx in grp])
        load_data_to_gpu(batch_dict, device)
        if args.infer_time:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        with torch.no_grad():
            pred_dicts, ret_dict = model(batch_dict)
        disp = {}
        if args.infer_time:
            e1.record()
            e1.synchronize()
            if it >= start_iter:
                meter.update(e0.elapsed_time(e1))
            disp["infer_time"] = "%.2f(%.2f)" % (meter.val, meter.avg)
        statistics_info(cfg, ret_dict, metric, disp)
        det_annos += dataset.generate_prediction_dicts(batch_dict, pred_dicts, class_names)
        if rank == 0 and (it % max(1, len(groups) // 10) == 0):
            logger.info("eval %d/%d %s" % (it + 1, len(groups), disp))
    sec_per_example = (time.time() - t_epoch) / max(1, len(indices))
    logger.info("Generate label finished(sec_per_example: %.4f second)." % sec_per_example)
    if world > 1:       # merge_results_dist (common_utils.py:229-250): rank files + barrier, interleaved back into frame order
        import torch.distributed as dist
        tmp = os.path.join(result_dir, "tmpdir")
        os.makedirs(tmp, exist_ok=True)
        with open(os.path.join(tmp, "result_part_%d.pkl" % rank), "wb") as f:
            pickle.dump((det_annos, metric), f)
        dist.barrier()
        if rank != 0:
            return {}, det_annos
        parts = [pickle.load(open(os.path.join(tmp, "result_part_%d.pkl" % r), "rb")) for r in range(world)]
        merged = []
        for i in range(max(len(p[0]) for p in parts)):
            for p in parts:
                if i < len(p[0]):
                    merged.append(p[0][i])
        det_annos = merged
        metric = {k: sum(p[1][k] for p in parts) for k in metric}
    ret = {}
    gt = metric["gt_num"]
    for t in cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST:
        ret["recall/roi_%s" % str(t)] = metric["recall_roi_%s" % str(t)] / max(gt, 1)
        ret["recall/rcnn_%s" % str(t)] = metric["recall_rcnn_%s" % str(t)] / max(gt, 1)
        logger.info("recall_roi_%s: %f" % (t, ret["recall/roi_%s" % str(t)]))
        logger.info("recall_rcnn_%s: %f" % (t, ret["recall/rcnn_%s" % str(t)]))
    total = sum(len(a["name"]) for a in det_annos)
    logger.info("Average predicted number of objects(%d samples): %.3f" % (len(det_annos), total / max(1, len(det_annos))))
    ret["avg_pred_objects"] = total / max(1, len(det_annos))
    if args.infer_time:
        ret["infer_time_ms"] = meter.avg
        logger.info("infer_time (HIP events, after a 10 %% warm-up): %.3f ms / frame" % meter.avg)
    with open(os.path.join(result_dir, "result.pkl"), "wb") as f:
        pickle.dump(det_annos, f)
    logger.info("Result is saved to %s" % result_dir)
    return ret, det_annos


def parse_config(argv=None):
    ap = argparse.ArgumentParser(description="arg parser")
    ap.add_argument("--cfg_file", type=str, required=True)
    ap.add_argument("--ckpt", type=str, default=None, help="pcdet checkpoint ({'model_state': ...}); default: seeded synthetic weights")
    ap.add_argument("--frames", type=str, default=None, help="root directory of pre-extracted frames (Waymo .npy / nuScenes .bin)")
    ap.add_argument("--infos", type=str, default=None, help="info pickle of the reference's data preparation (with --frames)")
    ap.add_argument("--kind", default="waymo", choices=["waymo", "nuscenes"], help="frame format (with --frames)")
    ap.add_argument("--synthetic", type=int, default=8, help="number of seeded synthetic frames when --frames is not given")
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--voxels", type=int, default=30000)
    ap.add_argument("--batch_size", type=int, default=1, help="frames per forward (tools/test.py:25)")
    ap.add_argument("--quant", default="sq", choices=["sq", "plain", "none"])
    ap.add_argument("--w_bits", type=int, default=8)
    ap.add_argument("--act_bits", type=int, default=8)
    ap.add_argument("--alpha", type=float, default=0.5)
    ap.add_argument("--static", type=int, default=0, help="static PTQ: calibrate on this many frames first (quant/quantize.py:175-207)")
    ap.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    ap.add_argument("--infer_time", action="store_true", default=False)
    ap.add_argument("--result_dir", type=str, default="output/eval")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_config(argv)
    from amdq3d import dist as D
    from amdq3d import quant
    from amdq3d.config import AttrDict, cfg_from_yaml_file
    from amdq3d.datasets import FrameFileDataset, SyntheticDataset
    from amdq3d.detector import build_network
    from amdq3d.synthetic import calibrate_heat_map_on_device, init_synthetic_weights
    logging.basicConfig(level=logging.INFO, format="%(asctime)s  %(levelname)5s  %(message)s")
    logger = logging.getLogger("eval")
    cfg = cfg_from_yaml_file(os.path.join(ROOT, args.cfg_file) if not os.path.isabs(args.cfg_file) else args.cfg_file, AttrDict())
    world, rank, local = D.env_world() if args.launcher == "pytorch" else (1, 0, 0)
    if not torch.cuda.is_available():
        raise SystemExit("tools/eval.py needs an MI355X (the detector has no CPU path)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        D.init("nccl")
    if args.frames:
        if not args.infos:
            raise SystemExit("--frames needs --infos (the reference's info pickle)")
        with open(args.infos, "rb") as f:
            infos = pickle.load(f)
        dataset = FrameFileDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, args.frames, infos, args.kind)
    else:
        dataset = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_frames=args.synthetic, n_points=args.points, n_voxels=args.voxels)
    model = build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=dataset)
    if args.ckpt:
        model.load_params_from_file(filename=args.ckpt, logger=logger, to_cpu=True)
    else:
        logger.info("no --ckpt: seeded synthetic weights; heat-map statistics from one calibration frame run on the device")
        init_synthetic_weights(model, 4)
        model.eval()
        model.cuda()
        cal = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_frames=1, n_points=args.points, n_voxels=args.voxels, seed0=1004) \
            if not args.frames else dataset
        from amdq3d.datasets import load_data_to_gpu
        cb = cal.collate_batch([cal[0]])
        load_data_to_gpu(cb, device)
        calibrate_heat_map_on_device(model, cb)
    model.eval()
    if args.quant != "none":
        if cfg.MODEL.NAME == "VoxelNeXt":
            quant.dynamic_quant_voxelnext(model, args.w_bits, args.act_bits, sq=args.quant == "sq")
        else:
            calib = [dataset.collate_batch([dataset[i % len(dataset)]]) for i in range(args.static)] if args.static else None
            if args.static:
                model.cuda()        # the calibration forwards run on the device (quant_centerpoint.py:252)
            quant.quant(model, args.w_bits, args.act_bits, sq=args.quant == "sq", alpha=args.alpha, static=bool(args.static),
                        test_loader=calib, n_batches=args.static)
    model.cuda()
    indices = D.frames_for_rank(len(dataset), rank, world)
    ret, _ = eval_one_epoch(cfg, args, model, dataset, indices, logger, args.result_dir, device, rank, world)
    if rank == 0:
        logger.info("****************Evaluation done.*****************")
        print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in ret.items()})
    D.shutdown()
    return ret


if __name__ == "__main__":
    main()
