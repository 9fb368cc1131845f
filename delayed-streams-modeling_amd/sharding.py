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


class _DevicePtrView:
    """A raw device allocation as a uint8 array for torch.as_tensor (zero copy, `__cuda_array_interface__` v2)."""

    def __init__(self, ptr, nbytes, owner):
        self._owner = owner  # keeps the allocation alive as long as the view is
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def broadcast_tensor_(t, src, dist, chunk_bytes=1 << 30):
    """In-place broadcast of a flat uint8 tensor (device or host) in pieces of `chunk_bytes`: xGMI is point to point
    (7 links x ~153 GB/s per GPU), a ring broadcast is per-link bound, and 1 GiB pieces keep RCCL's ring pipelined
    without any 2^31-element limit in the way."""
    n = t.numel()
    for lo in range(0, n, chunk_bytes):
        dist.broadcast(t[lo:min(n, lo + chunk_bytes)], src)
    return t


def fan_out_engine(dsm, cfg, batch_size, lm_path, mimi_path, dist, device, src=0):
    """One engine per rank with ONE collective at load: rank `src` reads the safetensors and packs the
    weights into its device arena (dsm_asr_create); the arena bytes and the loader manifest travel over RCCL (xGMI) and
    the other ranks attach to them (dsm_asr_create_from_arena) — no file, no conversion, no host copy on their side.
    Returns (engine, stats) with stats = {"arena_bytes", "broadcast_ms", "ranks"}."""
    import time
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    eng, manifest = None, b""
    hdr = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        eng = dsm.AsrEngine(cfg, batch_size, lm_path, mimi_path, device_id=device.index or 0)
        ptr, nbytes, manifest = eng.weight_arena()
        hdr[0], hdr[1] = nbytes, len(manifest)
    dist.broadcast(hdr, src)
    nbytes, mlen = int(hdr[0].item()), int(hdr[1].item())
    mt = torch.frombuffer(bytearray(manifest), dtype=torch.uint8).to(device) if rank == src and mlen else torch.zeros(max(mlen, 1), dtype=torch.uint8, device=device)
    dist.broadcast(mt, src)
    if rank == src:
        arena = torch.as_tensor(_DevicePtrView(ptr, nbytes, eng), device=device)
    else:
        arena = torch.empty(nbytes, dtype=torch.uint8, device=device)
    torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.perf_counter()
    broadcast_tensor_(arena, src, dist)
    torch.cuda.synchronize(device)
    dist.barrier()
    ms = (time.perf_counter() - t0) * 1000.0
    if rank != src:
        manifest = bytes(mt[:mlen].cpu().numpy().tobytes())
        eng = dsm.AsrEngine(cfg, batch_size, device_id=device.index or 0, arena=(arena.data_ptr(), nbytes, manifest, arena))
    return eng, {"arena_bytes": nbytes, "broadcast_ms": ms, "ranks": world}


def digest(raw):
    return hashlib.sha256(np.ascontiguousarray(raw).tobytes()).hexdigest()


class WorkerRouter:
    """One process, several GPUs: one engine + one worker per device, a new socket goes to the FIRST worker with a free slot —
    `BatchedAsr::channels` (srv/batched_asr.rs:796-808: the first `None` of the slot table, "Server at capacity" when there is
    none) across workers.  Stream ids are (worker index, slot); slots never migrate (a stream's ring caches live on one device).
    Pure host logic: `workers` are dsm_amd.Worker objects (or anything with open / close_channel that raises at capacity)."""

    def __init__(self, workers):
        self.workers = list(workers)

    def open(self):
        for i, w in enumerate(self.workers):
            try:
                return i, w.open()
            except Exception as ex:  # this worker is full: try the next device
                if "capacity" not in str(ex):
                    raise
        raise RuntimeError("Server at capacity")

    def close(self, stream):
        i, slot = stream
        self.workers[i].close_channel(slot)

    def worker_of(self, stream):
        return self.workers[stream[0]]
