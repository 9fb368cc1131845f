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


@pytest.mark.parametrize("groups", ["3", "1"])
def test_stream_groups_do_not_change_results(gpu, dsm, lib, orc, tiny_weights, monkeypatch, groups):
    """The LM step splits the batch into stream groups that free-run on their own HIP streams (slots 0-15, 16-31,
    32-39 here).  Rows are independent, so tokens, VAD heads and words must equal the oracle's for any grouping,
    through masks, resets and the device-pointer entry points that never join the groups."""
    import torch
    from dsm_amd import synth
    monkeypatch.setenv("DSM_LM_GROUPS", groups)
    cfg = dsm.config_tiny()
    B, steps = 40, 16
    eng = dsm.AsrEngine(cfg, B, *tiny_weights)
    ora = orc.OracleAsr(cfg, B, *tiny_weights)
    pcm = synth.synth_pcm(B, steps)
    rng = np.random.default_rng(4)
    masks = (rng.random((steps, B)) < 0.8).astype(np.uint8)
    nh = cfg.extra_heads_num
    dev = torch.device("cuda", 0)
    d_text = torch.zeros(B, dtype=torch.int32, device=dev)
    d_prs = torch.zeros(nh * B, dtype=torch.float32, device=dev)
    d_codes = torch.zeros(B * cfg.audio_codebooks, dtype=torch.int32, device=dev)
    for s in range(steps):
        if s == 7:
            for slot in (3, 20, 39):
                eng.reset_batch_idx(slot); ora.reset_batch_idx(slot)
        act = masks[s].astype(bool)
        oc, ot, op = ora.step_pcm(pcm[s], masks[s])
        if s % 2 == 0:  # host-pointer entry
            ec, et, ep = eng.step_pcm(pcm[s], masks[s])
        else:  # device-pointer entry: nothing synchronised, groups keep running into the next call
            d_pcm = torch.from_numpy(pcm[s]).to(dev)
            d_mask = torch.from_numpy(masks[s]).to(dev)
            torch.cuda.synchronize()  # torch's stream is not ordered against the engine's non-blocking streams
            eng.step_pcm_dev(d_pcm.data_ptr(), d_mask.data_ptr(), d_codes.data_ptr(), d_text.data_ptr(), d_prs.data_ptr())
            eng.sync()
            ec = d_codes.cpu().numpy().astype(np.uint32).reshape(B, -1)
            et = d_text.cpu().numpy().astype(np.uint32)
            ep = d_prs.cpu().numpy().reshape(nh, B)
        assert np.array_equal(ec[act], oc[act]), f"codes differ at step {s}"
        assert np.array_equal(et[act], ot[act]), f"text tokens differ at step {s}"
        assert np.array_equal(ep[:, act].view(np.uint32), op[:, act].view(np.uint32)), f"VAD heads differ at step {s}"
    eng.close()
    ora.close()


def test_in_workgroup_chunk_loop_on_tiny_shapes(gpu, dsm, lib, orc, tiny_weights, monkeypatch):
    """DSM_CHUNK_LOOP_MIN=1 sends every multi-chunk GEMM of the tiny model (K = 352: one full and one 3-block chunk)
    down the in-workgroup chunk loop; tokens, codes and decoded PCM must not move."""
    from dsm_amd import synth
    monkeypatch.setenv("DSM_CHUNK_LOOP_MIN", "1")
    cfg = dsm.config_tiny()
    B = 5
    eng = dsm.AsrEngine(cfg, B, *tiny_weights)
    ora = orc.OracleAsr(cfg, B, *tiny_weights)
    pcm = synth.synth_pcm(B, 8)
    rng = np.random.default_rng(9)
    for s in range(8):
        mask = (rng.random(B) < 0.8).astype(np.uint8)
        act = mask.astype(bool)
        ec, et, ep = eng.step_pcm(pcm[s], mask)
        oc, ot, op = ora.step_pcm(pcm[s], mask)
        assert np.array_equal(ec[act], oc[act]) and np.array_equal(et[act], ot[act])
        assert np.array_equal(ep[:, act].view(np.uint32), op[:, act].view(np.uint32))
        lg_e = eng.debug_read("lm.logits", B * cfg.text_out_vocab_size).reshape(B, -1)
        lg_o = ora.debug_read("lm.logits", B * cfg.text_out_vocab_size).reshape(B, -1)
        assert np.array_equal(lg_e[act].view(np.uint32), lg_o[act].view(np.uint32))
    eng.close()
    ora.close()


@pytest.mark.parametrize("depth", ["4", "2"])
def test_whole_k_loop_kernel_on_small_models(gpu, dsm, lib, orc, tiny_weights, monkeypatch, depth):
    """DSM_CHUNK_LOOP_MIN=1 sends every split-K GEMM through gemm_loop_kernel (whole K inside the workgroup, rolling
    load window of 4 or 2 blocks — the large-batch path) on the tiny model and on a medium one whose gating width
    (1408 = 5 chunks + 4 blocks) ends in a partial chunk.  Same bits as the oracle, masks and a reset included."""
    import os
    from dsm_amd import synth
    from test_parity_gpu import run_pair
    monkeypatch.setenv("DSM_CHUNK_LOOP_MIN", "1")
    monkeypatch.setenv("DSM_LOOP_DEPTH", depth)
    rng = np.random.default_rng(2)
    masks = (rng.random((14, 4)) < 0.8).astype(np.uint8)
    masks[:, 0] = 1
    run_pair(dsm, orc, dsm.config_tiny(), 4, *tiny_weights, steps=14, mask_fn=lambda s: masks[s], resets={6: [2]})
    cfg = dsm.config_medium()
    lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="medium_bf16_hd128_ctx300")
    run_pair(dsm, orc, cfg, 4, lm, mimi, steps=10, mask_fn=lambda s: masks[s], resets={6: [2]})


@pytest.mark.parametrize("mt", ["4", "2", "1"])
def test_one_chunk_gemms_through_the_loop_kernel(gpu, dsm, lib, orc, tiny_weights, monkeypatch, mt):
    """One-chunk GEMMs (K <= 256: the first SEANet layers, the RVQ projections) take gemm_loop_kernel instead of
    gemm_tile_kernel once they span >= 1024 m-tiles (B >= 35 on real Mimi) — a size no oracle run reaches.  DSM_SMALLK_MIN=1
    sends every such GEMM of the tiny and the medium model down that path, with each tile height: same bits as the oracle."""
    import os
    from dsm_amd import synth
    from test_parity_gpu import run_pair
    monkeypatch.setenv("DSM_SMALLK_MIN", "1")
    monkeypatch.setenv("DSM_SMALLK_MT", mt)
    rng = np.random.default_rng(4)
    masks = (rng.random((10, 5)) < 0.8).astype(np.uint8)
    masks[:, 0] = 1
    run_pair(dsm, orc, dsm.config_tiny(), 5, *tiny_weights, steps=10, mask_fn=lambda s: masks[s], resets={4: [1]})
    cfg = dsm.config_medium()
    lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="medium_bf16_hd128_ctx300")
    run_pair(dsm, orc, cfg, 5, lm, mimi, steps=6, mask_fn=lambda s: masks[s], resets={4: [1]})


def test_wide_model_d4096(gpu, dsm, lib, orc):
    """d_model = 4096 is the widest row the row kernels take (`DSM_ROW_ITS` = 4 float4 per thread) and the only width
    that reaches `gemm_reduce_rows_kernel<4>`; every shipped model stops at 2048, so this one-layer LM (32 heads of 128,
    gating hidden 11264, 16 K-chunks) is what runs them — with the fused QKV prologue over its split-K slabs, through
    ring wrap on a 12-frame ring, bit for bit (paths no test reached)."""
    import os
    from dsm_amd import synth
    from test_parity_gpu import run_pair
    cfg = dsm.config_medium(lm_heads=32, lm_head_dim=128, lm_context=12, lm_layers=1, mimi_context=10)
    assert cfg.lm.d_model == 4096
    lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="medium_d4096")
    B = 3
    masks = np.ones((16, B), dtype=np.uint8)
    masks[5:9, 1] = 0
    run_pair(dsm, orc, cfg, B, lm, mimi, steps=16, mask_fn=lambda s: masks[s], resets={10: [2]})
