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


def _exchange_worker(rank, world, port, every, S, frames, q):
    """Each rank drives the real double-buffered sequence of bench.py (vo.sharding.RecordExchange: post ... post -> join
    -> all-gather -> flip) with CPU tensors standing in for the device records and the collective asynchronous, as on the
    side stream: every record carries (rank, frame, sequence) and must come out of the gather it went into, intact, in
    order, exactly once -- a buffer that were written again before its gather had read it would show a later frame."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vo import sharding
    cap = 8
    L = sharding.record_length(cap)
    bufs = [torch.zeros(every * S * L, dtype=torch.float64) for _ in range(2)]
    outs = [torch.zeros(world * every * S * L, dtype=torch.float64) for _ in range(2)]
    seen = []

    def record(r, frame, seq):
        T = np.eye(4)
        T[0, 3], T[1, 3], T[2, 3] = r, frame, seq
        lm = np.full((1 + (frame + seq + r) % cap, 3), 1000.0 * r + 10.0 * frame + seq)
        return sharding.pack_record(T, lm, cap)

    def post(result, seq, buf, off):
        buf[off:off + L] = torch.from_numpy(record(rank, result, seq))

    def on_gathered(dst, n):
        rows = dst.numpy().reshape(world, every * S, L)
        for r in range(world):
            for k in range(n):
                (T, lm), = sharding.unpack_records(rows[r, k], 1, cap)
                seen.append((r, int(T[1, 3]), int(T[2, 3]), bool(np.array_equal(rows[r, k], record(r, int(T[1, 3]), int(T[2, 3]))))))

    x = sharding.RecordExchange(bufs, outs, L, every, S, post=post,
                                gather=lambda src, dst: dist.all_gather_into_tensor(dst, src, async_op=True),
                                on_gathered=on_gathered)
    for frame in range(frames):
        x.post([frame] * S)               # (the "StepResult" of every sequence: here just the frame number)
    x.finish()
    want = [(r, f, s_, True) for f0 in range(0, frames, every) for r in range(world)
            for f in range(f0, min(f0 + every, frames)) for s_ in range(S)]
    ok = seen == want and x.collectives == -(-frames // every)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _run_two_ranks(target, args):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(results)


def test_double_buffered_exchange_one_collective_per_frame():
    """one all-gather per frame step (every = 1), two sequences per rank."""
    assert _run_two_ranks(_exchange_worker, (1, 2, 9)) == [(0, True), (1, True)]


def test_double_buffered_exchange_batched_with_a_partial_last_batch():
    """bench.py's default shape in small: several frames per all-gather, the run ending inside a batch."""
    assert _run_two_ranks(_exchange_worker, (4, 3, 10)) == [(0, True), (1, True)]
