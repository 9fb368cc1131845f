"""N > 1 path on CPU (gloo, world_size 2): checkpoint broadcast and stream sharding.  The per-step path has
no collective — the property that makes that legal is checked here too: a stream's outputs do not depend on
which batch/rank it is served by."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_streams_and_route(dsm):
    from dsm_amd import sharding
    sh = sharding.shard_streams(512, 8)
    assert sh == [(64 * r, 64 * (r + 1)) for r in range(8)]
    sh = sharding.shard_streams(10, 4)
    assert [hi - lo for lo, hi in sh] == [3, 3, 2, 2] and sh[0][0] == 0 and sh[-1][1] == 10
    assert sharding.route(0, sh) == (0, 0) and sharding.route(9, sh) == (3, 1) and sharding.route(3, sh) == (1, 0)


WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
    import torch.distributed as dist
    import dsm_amd, oracle
    from dsm_amd import synth, sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = dsm_amd.config_tiny()
    wdir = {wdir!r}
    # only rank 0 holds the checkpoint; everyone else receives it through the collective
    paths = []
    for which in ("lm", "mimi"):
        raw = np.fromfile(os.path.join(wdir, "tiny.%s.safetensors" % which), dtype=np.uint8) if rank == 0 else np.zeros(0, np.uint8)
        got = sharding.broadcast_bytes(raw, 0, dist, chunk_bytes=300000)  # several pieces (the files are ~1 MB)
        p = os.path.join(wdir, "shard_test.rank%d.%s.safetensors" % (rank, which))
        got.tofile(p)
        paths.append(p)
    dig = [sharding.digest(np.fromfile(p, dtype=np.uint8)) for p in paths]
    n_streams, steps = 4, 5
    lo, hi = sharding.shard_streams(n_streams, world)[rank]
    pcm = synth.synth_pcm(n_streams, steps)[:, lo:hi]
    o = oracle.OracleAsr(cfg, hi - lo, *paths)
    out = []
    for s in range(steps):
        mask = np.ones(hi - lo, dtype=np.uint8)
        codes, text, prs = o.step_pcm(np.ascontiguousarray(pcm[s]), mask)
        out.append([codes.tolist(), text.tolist()])
    for p in paths:
        os.remove(p)
    json.dump({{"digest": dig, "lo": lo, "hi": hi, "out": out}}, open(os.path.join(wdir, "shard_test.rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_broadcast_and_shard_independence(dsm, orc, tiny_weights, tmp_path):
    import json
    from dsm_amd import synth, sharding
    wdir = os.path.dirname(tiny_weights[0])
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, wdir=wdir))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)], env=env, timeout=600)
    res = [json.load(open(os.path.join(wdir, f"shard_test.rank{r}.json"))) for r in range(2)]
    want_dig = [sharding.digest(np.fromfile(p, dtype=np.uint8)) for p in tiny_weights]
    assert res[0]["digest"] == want_dig and res[1]["digest"] == want_dig
    # one process serving all 4 streams must give every stream exactly what its shard gave it
    cfg = dsm.config_tiny()
    o = orc.OracleAsr(cfg, 4, *tiny_weights)
    pcm = synth.synth_pcm(4, 5)
    for s in range(5):
        codes, text, _ = o.step_pcm(pcm[s], np.ones(4, dtype=np.uint8))
        for r in res:
            lo, hi = r["lo"], r["hi"]
            assert codes[lo:hi].tolist() == r["out"][s][0] and text[lo:hi].tolist() == r["out"][s][1]


def test_router_sends_a_new_socket_to_the_first_worker_with_a_free_slot():
    """BatchedAsr::channels across workers (srv/batched_asr.rs:796-808), the host half of the one-process multi-GPU layout."""
    import pytest
    from dsm_amd.sharding import WorkerRouter

    class FakeWorker:
        def __init__(self, B):
            self.free = list(range(B))
        def open(self):
            if not self.free:
                raise RuntimeError("Server at capacity")
            return self.free.pop(0)
        def close_channel(self, slot):
            self.free.append(slot); self.free.sort()

    r = WorkerRouter([FakeWorker(2), FakeWorker(1), FakeWorker(2)])
    got = [r.open() for _ in range(5)]
    assert got == [(0, 0), (0, 1), (1, 0), (2, 0), (2, 1)]
    with pytest.raises(RuntimeError, match="Server at capacity"):
        r.open()
    r.close((1, 0)); r.close((0, 1))
    assert r.open() == (0, 1) and r.open() == (1, 0)  # the first free slot of the first worker that has one
