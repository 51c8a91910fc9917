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


class RecordExchange:
    """The exchange of the per-frame records, double-buffered: the records of `every` frames x `sequences` sequences are
    written back to back into batch buffer b (`post`); when the buffer is full -- or at `flush` -- the producer and the
    collective are ordered (`join`), ONE all-gather ships the buffer to every rank (`gather`) and the next records go to
    the other buffer.  `every` = 1 is SURVEY.md 8e's one collective per frame; larger values send fewer, larger messages
    (issuing a collective costs the host about a third of a one-sequence step).

    A buffer is written again only after the all-gather that reads it has been ordered behind: on the GPU through
    stream order (`join` = vo_pipeline_export_state_join: later records are written after everything the collective's
    stream held), with CPU tensors by waiting for the gather's work handle.  The device and the CPU form are this one
    class; the callables differ:
        post(result, q, buffer, offset)   write sequence q's record of `result` at buffer[offset : offset + rec_len]
        join()                            order producer and collective both ways (None: nothing to do)
        gather(src, dst)                  all-gather src into dst; may return a handle with .wait()
        on_gathered(dst, n_records)       optional: called once the batch in dst is complete on this rank
    """

    def __init__(self, buffers, gathered, rec_len, every, sequences, post, gather, join=None, on_gathered=None):
        assert len(buffers) == 2 and len(gathered) == 2
        self.buffers, self.gathered = buffers, gathered
        self.rec_len, self.every, self.S = int(rec_len), int(every), int(sequences)
        self.post_fn, self.gather_fn, self.join_fn, self.on_gathered = post, gather, join, on_gathered
        self.fill, self.buf = 0, 0
        self.pending = [None, None]          # per buffer: (handle or None, records in flight)
        self.collectives = 0

    def _settle(self, b):
        """The collective that last read buffer b (and wrote gathered[b]) is complete on the host's side."""
        if self.pending[b] is not None:
            handle, n = self.pending[b]
            if handle is not None:
                handle.wait()
            self.pending[b] = None
            if self.on_gathered is not None:
                self.on_gathered(self.gathered[b], n)

    def post(self, results):
        """The records of one collected frame (one StepResult per sequence)."""
        if self.fill == 0:
            self._settle(self.buf)           # (the gather that read this buffer two batches ago)
        for q in range(self.S):
            self.post_fn(results[q], q, self.buffers[self.buf], self.fill * self.rec_len)
            self.fill += 1
        if self.fill == self.every * self.S:
            self.flush()

    def flush(self):
        """Ships what has been posted so far (a partial batch travels as a whole buffer: fixed message size)."""
        if self.fill == 0:
            return
        if self.join_fn is not None:
            self.join_fn()
        handle = self.gather_fn(self.buffers[self.buf], self.gathered[self.buf])
        self.pending[self.buf] = (handle, self.fill)
        self.collectives += 1
        self.buf ^= 1
        self.fill = 0

    def finish(self):
        self.flush()
        for b in (self.buf, self.buf ^ 1):
            self._settle(b)
