"""One process per GPU; frames never cross GPUs. The only communication on this path is the throughput-aggregation
window: a barrier on each side of the timed region and one tiny all-reduce. On ROCm
`backend='nccl'` IS RCCL (xGMI inside a node); `gloo` is used by the CPU tests.

Replaces the reference's init_dist_pytorch / merge helpers (pcdet/utils/common_utils.py:189-204,229-250,
pcdet/datasets/__init__.py:31-51 for the frame striding)."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment. Returns (world, rank, local_rank)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    return world, rank, local


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def frames_for_rank(n_frames, rank, world):
    """Frame i goes to rank i mod world (the reference's non-shuffled DistributedSampler striding)."""
    return list(range(rank, n_frames, world))


def aggregate_throughput(frames_done, elapsed_s, device=None):
    """Whole-job frames/s: sum of frames over ranks / max elapsed over ranks. One 2-element all-reduce pair."""
    if not (dist.is_available() and dist.is_initialized()):
        return frames_done, elapsed_s, frames_done / elapsed_s if elapsed_s > 0 else 0.0
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                             if dist.get_backend() == "nccl" else torch.device("cpu"))
    f = torch.tensor([float(frames_done)], dtype=torch.float64, device=dev)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=dev)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    total, tmax = float(f.item()), float(t.item())
    return total, tmax, total / tmax if tmax > 0 else 0.0


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
