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


def test_ticketed_backend_whose_encode_fails_once(dsm, lib, orc, tiny_weights):
    """an encode_async that fails must cost one frame, not the pipeline.  A scripted ticket backend (the oracle
    behind a four-entry ring, like the engine's) refuses one call: the worker reports the backend's error for that pass, and
    every later frame still goes through with its own ticket — against a second worker whose backend never fails and that
    is simply not sent the lost frame."""
    import ctypes as C
    cfg = dsm.config_tiny()
    B, FRAME = 2, 1920

    def make(fail_at):
        ora = orc.OracleAsr(cfg, B, *tiny_weights)
        be, keep = worker_ref.oracle_backend(dsm, ora, cfg, B)
        ring = {"codes": {}, "next": 0, "calls": 0, "in_flight": set()}

        def encode_async(_s, pcm, mask, ticket):
            ring["calls"] += 1
            if ring["calls"] == fail_at:
                return -4  # DSM_ERR_STATE: nothing taken from the ring
            t = ring["next"]
            assert t not in ring["in_flight"], "the worker ran ahead of the ring"
            p = np.ctypeslib.as_array(pcm, shape=(B, FRAME)).copy()
            m = np.ctypeslib.as_array(mask, shape=(B,)).copy()
            ring["codes"][t] = np.where(m[:, None].astype(bool), ora.encode_step(p, m, side=0), 0).astype(np.uint32)
            ring["in_flight"].add(t)
            ring["next"] = (t + 1) % 4
            ticket[0] = t
            return 0

        def step_ticket(_s, t, mask, text, prs):
            assert t in ring["in_flight"], f"ticket {t} was never handed out"
            ring["in_flight"].discard(t)
            m = np.ctypeslib.as_array(mask, shape=(B,)).copy()
            tt, p = ora.step_tokens(ring["codes"].pop(t), m)
            np.ctypeslib.as_array(text, shape=(B,))[:] = tt
            if cfg.extra_heads_num:
                np.ctypeslib.as_array(prs, shape=(cfg.extra_heads_num, B))[:] = p
            return 0

        cbs = (dsm.BE_ENCODE_ASYNC(encode_async), dsm.BE_STEP_TICKET(step_ticket))
        be.encode_async, be.step_ticket = cbs
        return dsm.Worker(backend=be), (keep, cbs, ora), ring

    wa, keep_a, ring_a = make(fail_at=4)
    wb, keep_b, ring_b = make(fail_at=-1)
    sa, sb = [wa.open() for _ in range(B)], [wb.open() for _ in range(B)]
    assert sa == sb
    rng = np.random.default_rng(11)
    failed = 0
    for it in range(12):
        frames = [(0.1 * rng.standard_normal(FRAME)).astype(np.float32) for _ in range(B)]
        for slot, f in zip(sa, frames):
            wa.send(slot, dsm.encode_in_msg("Audio", pcm=f))
        try:
            produced = wa.step_encode()
        except dsm.DsmError:
            failed += 1
            produced = None  # this frame is gone (the reference's encoder thread would have died with it: srv/utils.rs:376-384)
        if produced is None:
            continue
        for slot, f in zip(sb, frames):
            wb.send(slot, dsm.encode_in_msg("Audio", pcm=f))
        assert wb.step_encode() == produced
        assert wa.step_model() == wb.step_model()
        for slot in sa:
            assert wa.recv(slot) == wb.recv(slot)
    assert failed == 1 and not ring_a["in_flight"] and not ring_b["in_flight"]
    wa.close(); wb.close()


def test_send_body_is_handle_query(dsm, lib, orc, tiny_weights):
    """BatchedAsr::handle_query (srv/batched_asr.rs:811-851): an mp3 file as the request body = Audio { pcm_decode + resample to
    24 kHz }, Marker { id: 0 }, ten seconds of silence — the same OutMsgs, in the same order, as those messages sent one by one,
    and the transcript ends when the Marker comes back."""
    import os
    cfg = dsm.config_tiny()
    B = 2
    body = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "audio", "loona.mp3"), "rb").read()
    pcm, rate, _ = dsm.mp3_decode(body)
    pcm24 = dsm.resample(pcm, rate, 24000)
    outs = []
    for via_body in (True, False):
        ora = orc.OracleAsr(cfg, B, *tiny_weights)
        be, keep = worker_ref.oracle_backend(dsm, ora, cfg, B)
        w = dsm.Worker(backend=be, detokenizer=lambda toks: "".join(chr(0x61 + t % 26) for t in toks))
        slot = w.open()
        if via_body:
            w.send_body(slot, body)
        else:
            assert w.send(slot, dsm.encode_in_msg("Audio", pcm=pcm24))
            assert w.send(slot, dsm.encode_in_msg("Marker", id=0))
            assert w.send(slot, dsm.encode_in_msg("Audio", pcm=np.zeros(240000, np.float32)))
        msgs = []
        for _ in range(400):
            w.step()
            got = w.recv(slot)
            msgs += got
            if any(m["type"] == "Marker" for m in got):
                break
        assert msgs[0] == {"type": "Ready"} and msgs[-1]["type"] == "Marker" or any(m["type"] == "Marker" for m in msgs)
        assert sum(m["type"] == "Step" for m in msgs) >= len(pcm24) // 1920
        outs.append(msgs)
        w.close()
    assert outs[0] == outs[1]
    with pytest.raises(dsm.DsmError):
        ora = orc.OracleAsr(cfg, B, *tiny_weights)
        be, keep = worker_ref.oracle_backend(dsm, ora, cfg, B)
        w = dsm.Worker(backend=be)
        w.send_body(w.open(), b"not an audio file at all" * 10)
