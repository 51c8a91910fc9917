"""Multi-GPU layout of the front-end: one process per GPU, one independent frame stream
per rank (a stream is a serial chain, so frames shard at sequence granularity), and one
all-gather per frame of every rank's fixed-size record

    [ T_cw 4x4 row-major (16) | n (1) | n landmarks x 3, n <= cap ]   float64

for the shared map.  The collective is torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU node, "gloo" in CPU tests).  At cap = 2000 a record is 48 KB: the
exchange is latency- not bandwidth-bound, so it is issued on a side stream and overlaps
the next frame's kernels (bench.py)."""
import numpy as np


def record_length(cap: int) -> int:
    return 17 + 3 * cap


def pack_record(T_cw: np.ndarray, landmarks: np.ndarray, cap: int) -> np.ndarray:
    rec = np.zeros(record_length(cap))
    rec[:16] = np.asarray(T_cw, dtype=np.float64).reshape(16)
    lm = np.asarray(landmarks, dtype=np.float64).reshape(-1, 3)[:cap]
    rec[16] = len(lm)
    rec[17:17 + 3 * len(lm)] = lm.reshape(-1)
    return rec


def unpack_records(gathered: np.ndarray, world: int, cap: int):
    """[(T_cw (4,4), landmarks (n,3))] for every rank."""
    out = []
    rows = np.asarray(gathered).reshape(world, record_length(cap))
    for row in rows:
        n = int(row[16])
        out.append((row[:16].reshape(4, 4).copy(), row[17:17 + 3 * n].reshape(n, 3).copy()))
    return out


def allgather_records(record, out=None):
    """All-gather one record tensor per rank into a (world * len) tensor (any backend)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * record.numel(), dtype=record.dtype, device=record.device)
    dist.all_gather_into_tensor(out, record)
    return out
