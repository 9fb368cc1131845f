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
    single-call worker delivers, in the same order (Step.buffered_pcm is the queue length when the frame was cut in both).
    Frames arrive one per pass, as from real-time clients: like the reference, a pass that finds several queued Audio messages
    for one channel keeps only the last complete frame (pre_process_pipelined drains the channel, :570-606)."""
    import threading
    import time
    import numpy as np
    cfg = dsm.config_tiny()
    B, frames = 4, 40
    rng = np.random.default_rng(9)
    audio = [[(0.1 * rng.standard_normal(1920 + int(rng.integers(0, 3)) * 64)).astype(np.float32) for _ in range(frames)] for _ in range(B)]

    def feed(w, slots, f):
        for slot in slots:
            w.send(slot, dsm.encode_in_msg("Audio", pcm=audio[slot][f]))
            if f % 9 == 4:
                w.send(slot, dsm.encode_in_msg("Marker", id=100 * slot + f))

    ea = dsm.AsrEngine(cfg, B, *tiny_weights)
    wa = dsm.Worker(ea)
    sa = [wa.open() for _ in range(B)]
    for f in range(frames):
        feed(wa, sa, f)
        assert wa.step()
    for _ in range(cfg.asr_delay_in_tokens + 8):  # let queued remainders and markers drain
        wa.step()
    want = {s: wa.recv(s) for s in sa}
    wa.close(); ea.close()

    eb = dsm.AsrEngine(cfg, B, *tiny_weights)
    wb = dsm.Worker(eb)
    sb = [wb.open() for _ in range(B)]
    assert sb == sa
    done = threading.Event()
    stats = {"max_ahead": 0, "errors": [], "encoded": 0, "stepped": 0}

    def encoder_loop():
        try:
            for f in range(frames):
                feed(wb, sb, f)
                while not wb.step_encode():  # queue full: the model side is three frames behind
                    time.sleep(0.0002)
                stats["encoded"] += 1
                stats["max_ahead"] = max(stats["max_ahead"], stats["encoded"] - stats["stepped"])
            for _ in range(cfg.asr_delay_in_tokens + 8):
                t0 = time.time()
                while not wb.step_encode() and time.time() - t0 < 0.05:
                    time.sleep(0.0002)
        except Exception as ex:  # pragma: no cover
            stats["errors"].append(ex)
        done.set()

    def model_loop():
        try:
            while True:
                if wb.step_model():
                    stats["stepped"] += 1
                elif done.is_set():
                    break
                else:
                    time.sleep(0.0001)
        except Exception as ex:  # pragma: no cover
            stats["errors"].append(ex)

    te, tm = threading.Thread(target=encoder_loop), threading.Thread(target=model_loop)
    te.start(); tm.start(); te.join(120); tm.join(120)
    assert not stats["errors"], stats["errors"]
    while wb.step_model():
        pass
    got = {s: wb.recv(s) for s in sb}
    for s in sa:
        assert sum(m["type"] == "Step" for m in want[s]) >= frames
        assert got[s] == want[s], f"slot {s}: pipelined worker delivered different messages"
    assert stats["max_ahead"] >= 1
    # both threads were inside the library while the encode / LM-group sequences were captured: the captures must have held
    # (capture_failures == 0) and the graphs must have been replayed (a silent eager fallback passed unnoticed)
    m = eb.metrics()
    assert m.capture_failures == 0, m.capture_error
    assert m.graph_launches > 0
    wb.close(); eb.close()
