"""CPU: the N > 1 path on gloo with 2 ranks — frame striding, barrier window, throughput aggregation."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update({"WORLD_SIZE": str(world), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from amdq3d import dist as D
    w, r, l = D.init("gloo")
    frames = D.frames_for_rank(8, r, w)
    D.barrier()
    # rank r "processes" its frames in (r + 1) seconds
    total, tmax, fps = D.aggregate_throughput(len(frames), float(r + 1))
    D.barrier()
    q.put((r, frames, total, tmax, fps))
    D.shutdown()


def test_two_rank_gloo_aggregation():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]        # frame i → rank i mod world, disjoint cover
    for r, frames, total, tmax, fps in res:
        assert total == 8.0 and tmax == 2.0 and fps == 4.0                 # sum of frames / max elapsed, same on every rank


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from amdq3d import dist as D
    assert D.frames_for_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.aggregate_throughput(10, 2.0) == (10, 2.0, 5.0)
    D.barrier()


def test_bench_gives_every_rank_its_own_frames():
    """bench.py: rank r works on frames seeded 4 + 8 r ... 4 + 8 r + 7 — N = 1 and N = 8 do the same per-GPU work, no frame twice."""
    sys.path.insert(0, ROOT)
    import bench
    seeds = [set(range(bench.rank_seed0(r), bench.rank_seed0(r) + bench.N_FRAMES)) for r in range(8)]
    assert seeds[0] == set(range(4, 12))
    assert all(len(s) == bench.N_FRAMES for s in seeds)
    assert len(set().union(*seeds)) == 8 * bench.N_FRAMES
