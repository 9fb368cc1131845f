"""The BatchedAsr worker on the HIP engine (dsm_worker_create) against the Python restatement of the reference's host
logic running the CPU oracle: same script of clients (ragged audio chunks, markers, a departure and a takeover of the
freed slot, a full house), same messages on every channel — as decoded from the msgpack bytes of the wire."""
import pytest

import worker_ref
from test_worker_cpu import run_script

pytestmark = pytest.mark.gpu


def test_worker_on_engine_matches_reference_logic(gpu, dsm, lib, orc, tiny_weights):
    cfg = dsm.config_tiny()
    B = 4
    detok = lambda toks: " ".join(f"<{t}>" for t in toks)
    eng = dsm.AsrEngine(cfg, B, *tiny_weights)
    worker = dsm.Worker(engine=eng, detokenizer=detok)
    ref = worker_ref.RefWorker(orc.OracleAsr(cfg, B, *tiny_weights), B, cfg.asr_delay_in_tokens, cfg.extra_heads_num, detok)
    run_script(dsm, worker, ref, worker_ref.script(B, 60, seed=8), B)
    worker.close()
    eng.close()


def test_two_thread_run_ahead_pipeline_delivers_the_same_messages(gpu, dsm, lib, tiny_weights):
    """The reference's thread pair (srv/batched_asr.rs:314 encoder_loop, :432 model_loop) on the engine: one thread keeps
    cutting and encoding frames (dsm_worker_step_encode, up to three ahead: dsm_mimi_encode_step_async), the other steps the
    LM and fans the messages out (dsm_worker_step_model: dsm_asr_step_tokens_ticket).  Every channel must receive what the
    single-call worker delivers, in the same order (Step.buffered_pcm is the queue length when the frame was cut in both)."""
    import threading
    import numpy as np
    cfg = dsm.config_tiny()
    B, frames = 4, 40
    rng = np.random.default_rng(9)
    audio = [[(0.1 * rng.standard_normal(1920)).astype(np.float32) for _ in range(frames)] for _ in range(B)]

    def feed(w, slots):
        for slot in slots:
            for f in range(frames):
                w.send(slot, dsm.encode_in_msg("Audio", pcm=audio[slot][f]))
                if f % 9 == 4:
                    w.send(slot, dsm.encode_in_msg("Marker", id=100 * slot + f))

    ea = dsm.AsrEngine(cfg, B, *tiny_weights)
    wa = dsm.Worker(ea)
    sa = [wa.open() for _ in range(B)]
    feed(wa, sa)
    while wa.step():
        pass
    want = {s: wa.recv(s) for s in sa}
    wa.close(); ea.close()

    eb = dsm.AsrEngine(cfg, B, *tiny_weights)
    wb = dsm.Worker(eb)
    sb = [wb.open() for _ in range(B)]
    assert sb == sa
    feed(wb, sb)
    done = threading.Event()
    stats = {"ahead": 0, "errors": []}

    def encoder_loop():
        try:
            idle = 0
            while idle < 200:
                if wb.step_encode():
                    idle = 0
                else:
                    idle += 1
                    done.wait(0.0005)
        except Exception as ex:  # pragma: no cover
            stats["errors"].append(ex)
        done.set()

    def model_loop():
        try:
            while not done.is_set() or wb.step_model():
                if not wb.step_model():
                    done.wait(0.0002)
        except Exception as ex:  # pragma: no cover
            stats["errors"].append(ex)

    te, tm = threading.Thread(target=encoder_loop), threading.Thread(target=model_loop)
    te.start(); tm.start(); te.join(120); tm.join(120)
    assert not stats["errors"], stats["errors"]
    while wb.step_model():
        pass
    got = {s: wb.recv(s) for s in sb}
    for s in sa:
        assert got[s] == want[s], f"slot {s}: pipelined worker delivered different messages"
        assert sum(m["type"] == "Step" for m in got[s]) == frames
    wb.close(); eb.close()
