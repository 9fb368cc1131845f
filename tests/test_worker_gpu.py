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
