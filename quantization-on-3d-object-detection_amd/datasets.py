"""Input side of the boundary: what the reference's dataset / dataloader hand to `model(batch_dict)`.

    DatasetTemplate attrs used by the detector    pcdet/datasets/dataset.py:26-40
    PointFeatureEncoder                           pcdet/datasets/processor/point_feature_encoder.py:4-57
    mask_points_by_range / shuffle_points         pcdet/utils/common_utils.py:78-81, processor/data_processor.py:79-105
    collate_batch (points branch)                 pcdet/datasets/dataset.py:220-325
    load_data_to_gpu                              pcdet/models/__init__.py:23-36

Voxelization moved onto the device (aq3d_voxelize), so `prepare_data` stops after the point-level steps and the
batch carries `points [N, 1+C]` (batch index in column 0) instead of `voxels / voxel_coords / voxel_num_points`.
"""
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
