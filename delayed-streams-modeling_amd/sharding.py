"""Multi-GPU plumbing: independent stream batches per rank, one collective at load.

Streams never interact (every op of the step is row-wise in the batch dimension:
core/batched_transformer.rs:74-121), so N GPUs run N engines on disjoint slot ranges and the only
data that crosses xGMI is the immutable checkpoint, broadcast once with RCCL (backend "nccl" on ROCm;
"gloo" in the CPU tests)."""
import hashlib

import numpy as np


def shard_streams(n_streams, world_size):
    """Contiguous slot ranges per rank: rank r serves global stream ids [lo, hi).  Sizes differ by at most one."""
    base, extra = divmod(n_streams, world_size)
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def route(stream_id, shards):
    """global stream id -> (rank, local slot)"""
    for r, (lo, hi) in enumerate(shards):
        if lo <= stream_id < hi:
            return r, stream_id - lo
    raise IndexError(stream_id)


def broadcast_bytes(raw, src, dist, device=None, chunk_bytes=512 << 20):
    """Broadcast a uint8 numpy array from rank `src` to every rank of the default process group: one collective for
    the length, then the payload in pieces of `chunk_bytes` (a 5 GB checkpoint stays far from any 2^31 element limit
    and needs only one chunk of device staging).  Returns the bytes as a numpy array on every rank."""
    import torch
    rank = dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([raw.size if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    total = int(n.item())
    out = np.ascontiguousarray(raw) if rank == src else np.empty(total, dtype=np.uint8)
    for lo in range(0, total, chunk_bytes):
        hi = min(total, lo + chunk_bytes)
        if rank == src:
            buf = torch.from_numpy(out[lo:hi]).to(dev)
        else:
            buf = torch.empty(hi - lo, dtype=torch.uint8, device=dev)
        dist.broadcast(buf, src)
        if rank != src:
            out[lo:hi] = buf.cpu().numpy()
        del buf
    return out


def digest(raw):
    return hashlib.sha256(np.ascontiguousarray(raw).tobytes()).hexdigest()
