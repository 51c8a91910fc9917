"""N > 1 path on CPU: world_size 2, gloo backend -- every rank runs its own sequence and
all-gathers its {pose, landmarks} record each frame (the collective bench.py issues over
RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vo import sharding, synthetic
    ok = True
    for frame in range(3):
        T = np.linalg.inv(synthetic.pose_world_cam(frame + 10 * rank))       # each rank: its own sequence
        rng = np.random.default_rng(100 * rank + frame)
        lm = rng.normal(size=(5 + rank + frame, 3))
        rec = torch.from_numpy(sharding.pack_record(T, lm, cap))
        allrec = sharding.allgather_records(rec).numpy()
        got = sharding.unpack_records(allrec, world, cap)
        for r in range(world):
            Tr = np.linalg.inv(synthetic.pose_world_cam(frame + 10 * r))
            lr = np.random.default_rng(100 * r + frame).normal(size=(5 + r + frame, 3))
            ok &= np.array_equal(got[r][0], Tr) and np.array_equal(got[r][1], lr)
    # max-over-ranks timing reduction as bench.py does it
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok &= float(t.item()) == float(world)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_rank_allgather_of_pose_and_landmark_records():
    world, cap = 2, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_record_roundtrip_and_capacity():
    from vo import sharding
    T = np.arange(16.0).reshape(4, 4)
    lm = np.arange(30.0).reshape(10, 3)
    rec = sharding.pack_record(T, lm, cap=4)                                  # truncated to the capacity
    assert rec.shape == (17 + 12,) and rec[16] == 4
    (T2, l2), = sharding.unpack_records(rec, 1, 4)
    assert np.array_equal(T2, T) and np.array_equal(l2, lm[:4])
