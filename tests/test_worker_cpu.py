"""The BatchedAsr host logic behind the C ABI (slot table, PCM queues, markers, fan-out, wire format), driven on the CPU
through dsm_worker_backend by the oracle, against the Python restatement in tests/worker_ref.py running a second
oracle instance: every channel must receive the same messages in the same order."""
import numpy as np
import pytest

import worker_ref


def run_script(dsm, worker, ref, events, B):
    open_slots = set()
    for step_events in events:
        for ev in step_events:
            if ev[0] == "open":
                a, b = worker.open(), ref.open()
                assert a == b
                open_slots.add(a)
            elif ev[0] == "open_expect_full":
                with pytest.raises(dsm.DsmError, match="Server at capacity"):
                    worker.open()
                with pytest.raises(RuntimeError):
                    ref.open()
            elif ev[0] == "close":
                worker.close_channel(ev[1]); ref.close_channel(ev[1])
                open_slots.discard(ev[1])
            elif ev[0] == "step" or ev[1] not in open_slots:
                continue
            elif ev[0] == "audio":
                assert worker.send(ev[1], dsm.encode_in_msg("Audio", pcm=ev[2]))
                ref.send(ev[1], {"type": "Audio", "pcm": ev[2]})
            elif ev[0] == "marker":
                assert worker.send(ev[1], dsm.encode_in_msg("Marker", id=ev[2]))
                ref.send(ev[1], {"type": "Marker", "id": ev[2]})
            elif ev[0] == "ping":
                assert worker.send(ev[1], dsm.encode_in_msg("Ping"))
            elif ev[0] == "garbage":
                assert worker.send(ev[1], b"\x93\x01\x02\x03") is False  # skipped, like the reference's warn + continue
        assert worker.step() == ref.step()
        for slot in range(B):
            got = worker.recv(slot) if slot in open_slots else []
            want = ref.recv(slot) if slot in open_slots and ref.ch[slot] is not None else []
            assert got == want, f"slot {slot}: {got} != {want}"
            if slot in open_slots:
                assert worker.buffered(slot) == ref.ch[slot]["data"].size


def test_worker_logic_on_the_oracle(dsm, lib, orc, tiny_weights):
    cfg = dsm.config_tiny()
    B = 4
    detok = lambda toks: "".join(chr(0x61 + t % 26) for t in toks)
    ora_c, ora_ref = orc.OracleAsr(cfg, B, *tiny_weights), orc.OracleAsr(cfg, B, *tiny_weights)
    be, keep = worker_ref.oracle_backend(dsm, ora_c, cfg, B)
    worker = dsm.Worker(backend=be, detokenizer=detok)
    ref = worker_ref.RefWorker(ora_ref, B, cfg.asr_delay_in_tokens, cfg.extra_heads_num, detok)
    events = worker_ref.script(B, 40)
    run_script(dsm, worker, ref, events, B)
    kinds = set()
    # the script must have exercised every message kind
    ref2 = worker_ref.RefWorker(orc.OracleAsr(cfg, B, *tiny_weights), B, cfg.asr_delay_in_tokens, cfg.extra_heads_num, detok)
    for step_events in events:
        for ev in step_events:
            if ev[0] == "open":
                ref2.open()
            elif ev[0] == "close":
                ref2.close_channel(ev[1])
            elif ev[0] == "audio" and ref2.ch[ev[1]] is not None:
                ref2.send(ev[1], {"type": "Audio", "pcm": ev[2]})
            elif ev[0] == "marker" and ref2.ch[ev[1]] is not None:
                ref2.send(ev[1], {"type": "Marker", "id": ev[2]})
        ref2.step()
        for slot in range(B):
            if ref2.ch[slot] is not None:
                kinds.update(m["type"] for m in ref2.recv(slot))
    assert {"Ready", "Step", "Marker", "Word", "EndWord"} <= kinds
    worker.close()


def test_idle_step_and_marker_without_audio(dsm, lib, orc, tiny_weights):
    """No data, no reset, no marker: the loop idles (encoder_loop sleeps, srv/batched_asr.rs:399).  A marker alone makes
    a step (all-inactive mask) and is released once the model step index reaches encoder step + delay (:583-593, :702-716)."""
    cfg = dsm.config_tiny()
    B = 2
    ora = orc.OracleAsr(cfg, B, *tiny_weights)
    be, keep = worker_ref.oracle_backend(dsm, ora, cfg, B)
    w = dsm.Worker(backend=be)
    assert w.step() is False
    slot = w.open()
    assert w.step() is True and w.recv(slot) == [{"type": "Ready"}]  # Init -> Ready + reset
    assert w.step() is False
    w.send(slot, dsm.encode_in_msg("Marker", id=9))
    assert w.step() is True and w.recv(slot) == []  # encoder step 1 -> due at model step 1 + delay
    frame = np.zeros(1920, dtype=np.float32)
    got = []
    for _ in range(cfg.asr_delay_in_tokens + 1):
        w.send(slot, dsm.encode_in_msg("Audio", pcm=frame))
        w.step()
        got.append([m["type"] for m in w.recv(slot)])
    # marker seen at encoder step 1 with nothing buffered -> due at step 1 + delay; the marker-only pass was model
    # step 2, audio pass k is model step 3 + k
    assert [k for ks in got for k in ks].count("Marker") == 1 and "Marker" in got[max(0, cfg.asr_delay_in_tokens - 2)]
    w.close()


def test_staged_calls_equal_the_single_call(dsm, lib, orc, tiny_weights):
    """dsm_worker_step_encode + dsm_worker_step_model (the reference's encoder_loop / model_loop split, srv/batched_asr.rs:314,432)
    against dsm_worker_step on a second worker: same messages per channel.  The oracle backend is synchronous (no tickets), so
    the encode stage refuses to run ahead of an unconsumed frame."""
    cfg = dsm.config_tiny()
    B = 3
    oa, ob = orc.OracleAsr(cfg, B, *tiny_weights), orc.OracleAsr(cfg, B, *tiny_weights)
    (bea, keepa), (beb, keepb) = worker_ref.oracle_backend(dsm, oa, cfg, B), worker_ref.oracle_backend(dsm, ob, cfg, B)
    wa, wb = dsm.Worker(backend=bea), dsm.Worker(backend=beb)
    rng = np.random.default_rng(5)
    slots = [wa.open() for _ in range(B)]
    assert slots == [wb.open() for _ in range(B)]
    for it in range(30):
        for slot in slots:
            if rng.random() < 0.8:
                pcm = (0.1 * rng.standard_normal(int(rng.integers(500, 4000)))).astype(np.float32)
                for w in (wa, wb):
                    w.send(slot, dsm.encode_in_msg("Audio", pcm=pcm))
            if rng.random() < 0.1:
                for w in (wa, wb):
                    w.send(slot, dsm.encode_in_msg("Marker", id=it))
        ran_b = wb.step()
        produced = wa.step_encode()
        assert produced == ran_b
        assert wa.step_encode() is False  # a synchronous backend holds one frame: no run-ahead
        assert wa.step_model() == ran_b and wa.step_model() is False
        for slot in slots:
            assert wa.recv(slot) == wb.recv(slot)
    wa.close(); wb.close()
