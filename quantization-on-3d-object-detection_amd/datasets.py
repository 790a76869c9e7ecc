"""Input side of the boundary: what the reference's dataset / dataloader hand to `model(batch_dict)`.

    DatasetTemplate attrs used by the detector    pcdet/datasets/dataset.py:26-40
    PointFeatureEncoder                           pcdet/datasets/processor/point_feature_encoder.py:4-57
    mask_points_by_range / shuffle_points         pcdet/utils/common_utils.py:78-81, processor/data_processor.py:79-105
    collate_batch (points branch)                 pcdet/datasets/dataset.py:220-325
    load_data_to_gpu                              pcdet/models/__init__.py:23-36

Voxelization moved onto the device (aq3d_voxelize), so `prepare_data` stops after the point-level steps and the
batch carries `points [N, 1+C]` (batch index in column 0) instead of `voxels / voxel_coords / voxel_num_points`.
"""
import os
from collections import defaultdict

import numpy as np
import torch

from .config import AttrDict


class PointFeatureEncoder:
    def __init__(self, config, point_cloud_range=None):
        self.point_encoding_config = config
        assert list(config.src_feature_list[0:3]) == ["x", "y", "z"]
        self.used_feature_list = list(config.used_feature_list)
        self.src_feature_list = list(config.src_feature_list)
        self.point_cloud_range = point_cloud_range

    @property
    def num_point_features(self):
        assert self.point_encoding_config.encoding_type == "absolute_coordinates_encoding"
        return len(self.used_feature_list)

    def forward(self, data_dict):
        points = data_dict["points"]
        cols = [points[:, 0:3]]
        for x in self.used_feature_list:
            if x in ("x", "y", "z"):
                continue
            idx = self.src_feature_list.index(x)
            cols.append(points[:, idx:idx + 1])
        data_dict["points"] = np.concatenate(cols, axis=1)
        data_dict["use_lead_xyz"] = True
        return data_dict


def mask_points_by_range(points, limit_range):
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) & \
           (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])


class DatasetTemplate(torch.utils.data.Dataset):
    """Attributes the detector reads at build time + the point-level half of prepare_data."""

    def __init__(self, dataset_cfg, class_names, training=False, logger=None):
        super().__init__()
        self.dataset_cfg = dataset_cfg
        self.training = training
        self.class_names = list(class_names)
        self.logger = logger
        self.point_cloud_range = np.array(dataset_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
        self.point_feature_encoder = PointFeatureEncoder(dataset_cfg.POINT_FEATURE_ENCODING, self.point_cloud_range)
        self.mode = "train" if training else "test"
        self.processors = {c.NAME: c for c in dataset_cfg.DATA_PROCESSOR}
        vcfg = self.processors["transform_points_to_voxels"]
        self.voxel_size = list(vcfg.VOXEL_SIZE)
        grid = (self.point_cloud_range[3:6].astype(np.float64) - self.point_cloud_range[0:3].astype(np.float64)) / \
            np.array(self.voxel_size, dtype=np.float64)
        self.grid_size = np.round(grid).astype(np.int64)
        self.depth_downsample_factor = None
        self.total_epochs = 0

    def prepare_data(self, data_dict):
        data_dict = self.point_feature_encoder.forward(data_dict)
        if "mask_points_and_boxes_outside_range" in self.processors:
            m = mask_points_by_range(data_dict["points"], self.point_cloud_range)
            data_dict["points"] = data_dict["points"][m]
        sh = self.processors.get("shuffle_points", None)
        if sh is not None and sh.SHUFFLE_ENABLED[self.mode]:
            idx = np.random.permutation(data_dict["points"].shape[0])
            data_dict["points"] = data_dict["points"][idx]
        return data_dict

    @staticmethod
    def collate_batch(batch_list, _unused=False):
        data_dict = defaultdict(list)
        for cur in batch_list:
            for k, v in cur.items():
                data_dict[k].append(v)
        batch_size = len(batch_list)
        ret = {}
        for key, val in data_dict.items():
            if key == "points":
                coors = [np.pad(c, ((0, 0), (1, 0)), mode="constant", constant_values=i) for i, c in enumerate(val)]
                ret[key] = np.concatenate(coors, axis=0)
                ret["points_batch_offsets"] = np.concatenate([[0], np.cumsum([len(c) for c in val])]).tolist()
            elif key in ("use_lead_xyz",):
                continue
            else:
                try:
                    ret[key] = np.stack(val, axis=0)
                except Exception:
                    ret[key] = val
        ret["batch_size"] = batch_size
        return ret

    @staticmethod
    def generate_prediction_dicts(batch_dict, pred_dicts, class_names, output_path=None):
        """pcdet/datasets/dataset.py:61-108 — numpy result records (the D2H boundary)."""
        annos = []
        for index, box_dict in enumerate(pred_dicts):
            scores = box_dict["pred_scores"].cpu().numpy()
            boxes = box_dict["pred_boxes"].cpu().numpy()
            labels = box_dict["pred_labels"].cpu().numpy()
            rec = {"name": np.array(class_names)[labels - 1] if len(labels) else np.zeros(0), "score": scores,
                   "boxes_lidar": boxes, "pred_labels": labels}
            if "frame_id" in batch_dict:
                rec["frame_id"] = batch_dict["frame_id"][index]
            annos.append(rec)
        return annos


def load_data_to_gpu(batch_dict, device=None):
    """pcdet/models/__init__.py:23-36: every ndarray except ids/metadata → float32 device tensor."""
    for key, val in batch_dict.items():
        if not isinstance(val, np.ndarray):
            continue
        if key in ("frame_id", "metadata", "calib", "image_paths", "ori_shape", "img_process_infos"):
            continue
        t = torch.from_numpy(val).float()
        batch_dict[key] = t.cuda() if device is None else t.to(device)
    return batch_dict


class SyntheticDataset(DatasetTemplate):
    """Seeded synthetic frames in the dataset's point-cloud range (pattern of tools/demo.py:23-60: raw point arrays in,
    prepare_data out). Frame i is generated from seed0 + i."""

    def __init__(self, dataset_cfg, class_names, n_frames=8, n_points=150000, n_voxels=30000, seed0=4, training=False,
                 logger=None, cache=True):
        super().__init__(dataset_cfg, class_names, training, logger)
        self.n_frames, self.n_points, self.n_voxels, self.seed0 = n_frames, n_points, n_voxels, seed0
        self._cache = {} if cache else None

    def __len__(self):
        return self.n_frames

    def raw_points(self, index):
        from .synthetic import make_frame
        if self._cache is not None and index in self._cache:
            return self._cache[index]
        pts = make_frame(self.seed0 + index, self.n_points, self.n_voxels, self.point_cloud_range, self.voxel_size,
                         self.grid_size, n_feat=len(self.point_feature_encoder.src_feature_list),
                         max_per_voxel=int(self.processors["transform_points_to_voxels"].MAX_POINTS_PER_VOXEL))
        if self._cache is not None:
            self._cache[index] = pts
        return pts

    def __getitem__(self, index):
        # the generator already shuffles with its own seeded stream; disable the global-RNG shuffle for determinism
        d = {"points": self.raw_points(index), "frame_id": "synthetic_%04d" % index}
        d = self.point_feature_encoder.forward(d)
        m = mask_points_by_range(d["points"], self.point_cloud_range)
        d["points"] = d["points"][m]
        return d


# ------------------------------------------------------------------ frame readers
def read_waymo_frame(path, disable_nlz_flag=False, tanh_dims=None):
    """One pre-extracted Waymo frame → float32 [N, 5] (x, y, z, intensity, elongation).

    Semantics of `WaymoDataset.get_lidar` (pcdet/datasets/waymo/waymo_dataset.py:196-208): the `.npy` rows are
    (x, y, z, intensity, elongation, NLZ flag); rows inside a no-label zone (flag != -1) are dropped unless
    DISABLE_NLZ_FLAG_ON_POINTS is set; intensity is squashed with tanh (or the dimensions listed in POINTS_TANH_DIM)."""
    raw = np.load(str(path))
    if raw.ndim != 2 or raw.shape[1] < 6:
        raise ValueError("%s: expected an [N, >=6] array (x, y, z, intensity, elongation, NLZ), got %s" % (path, raw.shape))
    pts = raw[:, 0:5].astype(np.float32, copy=True)
    if not disable_nlz_flag:
        pts = pts[raw[:, 5] == -1]
    for d in ([3] if tanh_dims is None else list(tanh_dims)):
        pts[:, d] = np.tanh(pts[:, d])
    return pts


def read_nuscenes_sweep(path, transform=None, time_lag=0.0, ego_radius=1.0, drop_ego=True):
    """One nuScenes `.bin` sweep (float32 rows of 5: x, y, z, intensity, ring) → ([N, 4] points, [N, 1] time lags).

    Semantics of `NuScenesDataset.get_sweep` (pcdet/datasets/nuscenes/nuscenes_dataset.py:86-100): points within
    `ego_radius` of the sensor in both x and y are the ego vehicle and are removed; `transform` (4x4, sweep → key-frame
    lidar) is applied to xyz in float64 and stored back as float32."""
    raw = np.fromfile(str(path), dtype=np.float32)
    if raw.size % 5 != 0:
        raise ValueError("%s: %d floats is not a whole number of 5-float points" % (path, raw.size))
    pts = raw.reshape(-1, 5)[:, :4].copy()
    if drop_ego:
        pts = pts[~((np.abs(pts[:, 0]) < ego_radius) & (np.abs(pts[:, 1]) < ego_radius))]
    if transform is not None:
        T = np.asarray(transform, dtype=np.float64)
        xyz1 = np.concatenate([pts[:, :3].astype(np.float64), np.ones((pts.shape[0], 1))], axis=1)
        pts[:, :3] = (xyz1 @ T.T)[:, :3].astype(np.float32)
    return pts, np.full((pts.shape[0], 1), float(time_lag))


def read_nuscenes_frame(root, info, max_sweeps=1, rng=None):
    """Key frame + (max_sweeps - 1) randomly chosen previous sweeps → float32 [N, 5] (x, y, z, intensity, time lag).

    Semantics of `NuScenesDataset.get_lidar_with_sweeps` (nuscenes_dataset.py:102-119): the key frame keeps its ego
    points and has time 0; the sweeps are drawn without replacement (np.random.choice there; pass `rng` for a
    reproducible draw), each through read_nuscenes_sweep with its `transform_matrix` and `time_lag`."""
    key, _ = read_nuscenes_sweep(os.path.join(str(root), info["lidar_path"]), drop_ego=False)
    parts, times = [key], [np.zeros((key.shape[0], 1))]
    sweeps = info.get("sweeps", [])
    n_extra = max_sweeps - 1
    if n_extra > 0:
        if n_extra > len(sweeps):
            raise ValueError("frame has %d sweeps, max_sweeps=%d needs %d" % (len(sweeps), max_sweeps, n_extra))
        pick = (rng or np.random).choice(len(sweeps), n_extra, replace=False)
        for k in pick:
            s = sweeps[int(k)]
            p, t = read_nuscenes_sweep(os.path.join(str(root), s["lidar_path"]), s.get("transform_matrix"), s.get("time_lag", 0.0))
            parts.append(p)
            times.append(t)
    pts = np.concatenate(parts, axis=0)
    return np.concatenate([pts, np.concatenate(times, axis=0).astype(pts.dtype)], axis=1)


class FrameFileDataset(DatasetTemplate):
    """Frames from files through the same prepare_data as the reference datasets: `infos` is the list of records of the
    reference's info pickles (Waymo: {'point_cloud': {'lidar_sequence', 'sample_idx'}, 'frame_id'}; nuScenes: {'lidar_path',
    'sweeps': [{'lidar_path', 'transform_matrix', 'time_lag'}], 'token'}); ground-truth boxes, when a record carries
    `gt_boxes` already in the unified lidar frame, ride along for the recall bookkeeping."""

    def __init__(self, dataset_cfg, class_names, root, infos, kind, training=False, logger=None, seed=None):
        super().__init__(dataset_cfg, class_names, training, logger)
        if kind not in ("waymo", "nuscenes"):
            raise ValueError("kind must be 'waymo' or 'nuscenes'")
        self.root, self.infos, self.kind = str(root), list(infos), kind
        self._rng = np.random.default_rng(seed) if seed is not None else None

    def __len__(self):
        return len(self.infos)

    def __getitem__(self, index):
        info = self.infos[index]
        if self.kind == "waymo":
            pc = info["point_cloud"]
            path = os.path.join(self.root, pc["lidar_sequence"], "%04d.npy" % pc["sample_idx"])
            pts = read_waymo_frame(path, bool(self.dataset_cfg.get("DISABLE_NLZ_FLAG_ON_POINTS", False)),
                                   self.dataset_cfg.get("POINTS_TANH_DIM", None))
            frame_id = info.get("frame_id", "%s_%03d" % (pc["lidar_sequence"], pc["sample_idx"]))
        else:
            pts = read_nuscenes_frame(self.root, info, int(self.dataset_cfg.get("MAX_SWEEPS", 1)), self._rng)
            frame_id = info.get("token", os.path.basename(info["lidar_path"]))
        d = {"points": pts, "frame_id": frame_id}
        if "gt_boxes" in info:
            d["gt_boxes"] = np.asarray(info["gt_boxes"], dtype=np.float32)
        return self.prepare_data(d)
