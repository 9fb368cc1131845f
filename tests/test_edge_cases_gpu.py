"""Edge cases of the boundary on the GPU: all-inactive steps, resets of idle slots, argument errors, repeated
create/destroy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_all_inactive_steps_freeze_every_state(gpu, dsm, lib, orc, tiny_weights):
    """A step with mask = 0 everywhere must leave conv carries, ring positions and item state untouched:
    the stream continues exactly as if the step had not happened (core/conv.rs:347-367, core/kv_cache.rs:150-154,
    core/asr.rs:221-223)."""
    from dsm_amd import synth
    cfg = dsm.config_tiny()
    B = 2
    a = dsm.AsrEngine(cfg, B, *tiny_weights)
    b = dsm.AsrEngine(cfg, B, *tiny_weights)
    pcm = synth.synth_pcm(B, 8)
    ones, zeros = np.ones(B, dtype=np.uint8), np.zeros(B, dtype=np.uint8)
    garbage = np.random.default_rng(0).standard_normal((B, 1920)).astype(np.float32)
    for s in range(8):
        if s in (2, 5):  # engine `a` sees two extra all-inactive frames of garbage
            a.step_pcm(garbage, zeros)
            assert all(m[0] == "Step" for m in a.poll_msgs())  # no Word/EndWord for inactive slots
        ca, ta, pa = a.step_pcm(pcm[s], ones)
        cb, tb, pb = b.step_pcm(pcm[s], ones)
        assert np.array_equal(ca, cb) and np.array_equal(ta, tb) and np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
    a.close()
    b.close()


def test_reset_of_idle_and_active_slots(gpu, dsm, lib, orc, tiny_weights):
    """Resetting a slot that never ran is a no-op; resetting slot 0 does not disturb slot 1 (core/asr.rs:257-266)."""
    from dsm_amd import synth
    cfg = dsm.config_tiny()
    B = 2
    eng = dsm.AsrEngine(cfg, B, *tiny_weights)
    ora = orc.OracleAsr(cfg, B, *tiny_weights)
    pcm = synth.synth_pcm(B, 6)
    mask = np.ones(B, dtype=np.uint8)
    for e in (eng, ora):
        e.reset_batch_idx(1)
    for s in range(6):
        if s == 3:
            eng.reset_batch_idx(0)
            ora.reset_batch_idx(0)
        ec, et, ep = eng.step_pcm(pcm[s], mask)
        oc, ot, op = ora.step_pcm(pcm[s], mask)
        assert np.array_equal(ec, oc) and np.array_equal(et, ot)
    eng.close()
    ora.close()


def test_argument_errors(gpu, dsm, lib, tiny_weights):
    cfg = dsm.config_tiny()
    eng = dsm.AsrEngine(cfg, 2, *tiny_weights)
    with pytest.raises(dsm.DsmError, match="out of range"):
        eng.reset_batch_idx(2)
    with pytest.raises(dsm.DsmError, match="out of range"):
        eng.mimi_reset_batch_idx(-1)
    assert lib.dsm_asr_step_tokens(eng.h, None, None, None, None) < 0
    assert lib.dsm_mimi_encode_step(eng.h, None, None, None, None) < 0
    eng.close()
    bad = dsm.config_tiny()
    bad.lm.num_heads = 3  # 128 / 3 is not an integer
    with pytest.raises(dsm.DsmError, match="d_model"):
        dsm.AsrEngine(bad, 2, *tiny_weights)
    bad = dsm.config_tiny()
    bad.lm.num_layers = 3  # the checkpoint only has 2 layers
    with pytest.raises(dsm.DsmError, match="cannot find tensor"):
        dsm.AsrEngine(bad, 2, *tiny_weights)
    bad = dsm.config_tiny()
    bad.text_out_vocab_size = 63
    with pytest.raises(dsm.DsmError, match="shape mismatch"):
        dsm.AsrEngine(bad, 2, *tiny_weights)


def test_create_destroy_cycles(gpu, dsm, lib, tiny_weights):
    import torch
    cfg = dsm.config_tiny()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        eng = dsm.AsrEngine(cfg, 8, *tiny_weights)
        eng.step_pcm(np.zeros((8, 1920), dtype=np.float32), np.ones(8, dtype=np.uint8))
        eng.decode_step(np.zeros((8, cfg.mimi.quantizer_n_q), dtype=np.uint32), np.ones(8, dtype=np.uint8))
        eng.close()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 << 20, "device memory leaked across create/destroy"
