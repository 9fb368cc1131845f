"""Multi-GPU load path on one GPU: the packed weight arena of a file-loaded engine is viewed as a torch
tensor without a copy, shipped (here: cloned; under torch.distributed.run with one rank: through the RCCL code path of
sharding.fan_out_engine), and a second engine attaches to the received bytes with dsm_asr_create_from_arena — no file,
no conversion.  Both engines must then give identical bits.  A truncated or foreign arena is refused."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(eng, cfg, B, pcm, steps):
    out = []
    for s in range(steps):
        mask = np.array([1] * B, dtype=np.uint8)
        mask[1] = s % 2
        codes, text, prs = eng.step_pcm(pcm[s], mask)
        hid = eng.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1)
        act = mask.astype(bool)
        out.append((codes[act].copy(), text[act].copy(), prs[:, act].copy(), hid[act].copy()))
    return out


def test_attach_to_a_copied_arena(gpu, dsm, lib, tiny_weights):
    import torch
    from dsm_amd import sharding, synth
    cfg = dsm.config_tiny()
    B = 3
    a = dsm.AsrEngine(cfg, B, *tiny_weights)
    ptr, nbytes, manifest = a.weight_arena()
    assert nbytes > 0 and nbytes % 256 == 0
    view = torch.as_tensor(sharding._DevicePtrView(ptr, nbytes, a), device=torch.device("cuda", 0))  # zero copy
    assert view.data_ptr() == ptr and view.numel() == nbytes and view.dtype == torch.uint8
    received = view.clone()
    torch.cuda.synchronize()
    b = dsm.AsrEngine(cfg, B, arena=(received.data_ptr(), nbytes, manifest, received))
    pcm = synth.synth_pcm(B, 6)
    ra, rb = _run(a, cfg, B, pcm, 6), _run(b, cfg, B, pcm, 6)
    for x, y in zip(ra, rb):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
        assert np.array_equal(x[2].view(np.uint32), y[2].view(np.uint32)) and np.array_equal(x[3].view(np.uint32), y[3].view(np.uint32))
    # decode side too (its weights live in the same arena)
    codes = np.zeros((B, cfg.mimi.quantizer_n_q), dtype=np.uint32)
    m = np.ones(B, dtype=np.uint8)
    assert np.array_equal(a.decode_step(codes, m).view(np.uint32), b.decode_step(codes, m).view(np.uint32))
    with pytest.raises(dsm.DsmError, match="arena"):
        dsm.AsrEngine(cfg, B, arena=(received.data_ptr(), nbytes - 256, manifest, received))
    other = dsm.config_tiny()
    other.lm.num_layers = 3
    with pytest.raises(dsm.DsmError):
        dsm.AsrEngine(other, B, arena=(received.data_ptr(), nbytes, manifest, received))
    b.close()
    a.close()


WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import torch, torch.distributed as dist
    import dsm_amd
    from dsm_amd import sharding, synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)   # RCCL, one rank: every collective of fan_out_engine still runs
    cfg = dsm_amd.config_tiny()
    lm, mimi = {lm!r}, {mimi!r}
    eng, stats = sharding.fan_out_engine(dsm_amd, cfg, 2, lm, mimi, dist, dev, src=0)
    ref = dsm_amd.AsrEngine(cfg, 2, lm, mimi)
    pcm = synth.synth_pcm(2, 4)
    same = True
    for s in range(4):
        m = np.ones(2, dtype=np.uint8)
        c1, t1, p1 = eng.step_pcm(pcm[s], m)
        c2, t2, p2 = ref.step_pcm(pcm[s], m)
        same = same and np.array_equal(c1, c2) and np.array_equal(t1, t2) and np.array_equal(p1.view(np.uint32), p2.view(np.uint32))
    print(json.dumps({{"same": bool(same), "stats": stats}}))
    eng.close(); ref.close()
    dist.barrier(); dist.destroy_process_group()
""")


def test_fan_out_engine_under_the_launcher(gpu, dsm, lib, tiny_weights, tmp_path):
    script = tmp_path / "fan.py"
    script.write_text(WORKER.format(root=ROOT, lm=tiny_weights[0], mimi=tiny_weights[1]))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29655", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["same"] is True and out["stats"]["ranks"] == 1 and out["stats"]["arena_bytes"] > 0


def test_replica_in_one_process_same_bits_and_owns_its_arena(gpu, dsm, lib):
    """dsm_asr_create_replica (r04: the one-process multi-device load behind the C ABI).  On the one-GPU box the replica lands on the
    source's own device (the copy is then a device-to-device memcpy instead of hipMemcpyPeerAsync; everything after it is the same
    code): a different batch size, the source destroyed FIRST (the replica owns its copy), and the same bits as the source for the
    same streams."""
    import numpy as np
    from dsm_amd import synth
    cfg = dsm.config_tiny()
    lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="tiny")
    src = dsm.AsrEngine(cfg, 3, lm, mimi)
    rep = dsm.AsrEngine.replica(src, 0, batch_size=5)
    assert rep.weight_arena()[0] != src.weight_arena()[0] and rep.weight_arena()[1] == src.weight_arena()[1]
    pcm = synth.synth_pcm(5, 6)
    want = []
    for s in range(6):
        want.append(src.step_pcm(pcm[s][:3], np.ones(3, np.uint8)))
    src.close()  # the replica must not depend on the source's memory
    for s in range(6):
        c, t, p = rep.step_pcm(pcm[s], np.ones(5, np.uint8))
        assert np.array_equal(c[:3], want[s][0]) and np.array_equal(t[:3], want[s][1])
        assert np.array_equal(p[:, :3].view(np.uint32), want[s][2].view(np.uint32))
    rep2 = dsm.AsrEngine.replica(rep, 0)  # a replica of a replica
    assert rep2.B == 5
    rep2.close(); rep.close()
    with pytest.raises(dsm.DsmError, match="no such device"):
        e = dsm.AsrEngine(cfg, 1, lm, mimi)
        try:
            dsm.AsrEngine.replica(e, 63)
        finally:
            e.close()
