"""config-toml surface (srv/main.rs:94-211): a BatchedAsr module table maps onto dsm_asr_config one to one."""
import ctypes as C
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _bytes(cfg):
    return bytes((C.c_char * C.sizeof(cfg)).from_buffer_copy(cfg))


def test_shipped_surface_matches_preset(dsm, lib):
    from dsm_amd import config_toml
    cfg, host = config_toml.load_batched_asr(os.path.join(HERE, "golden", "configs", "stt-en_fr.toml"))
    assert _bytes(cfg) == _bytes(dsm.config_stt_1b_en_fr())
    assert host["batch_size"] == 64 and host["module"] == "asr" and host["path"] == "/api/asr-streaming"
    assert host["lm_model_file"].startswith("hf://")


def test_rejects_what_the_batched_path_rejects(dsm, lib, tmp_path):
    from dsm_amd import config_toml
    src = open(os.path.join(HERE, "golden", "configs", "stt-en_fr.toml")).read()
    for old, new, msg in (("kv_repeat = 1", "kv_repeat = 2", "kv-repeat"),
                          ("use_conv_block = false", "use_conv_block = true", "conv-block"),
                          ("norm_first = true", "norm_first = false", "norm_first"),
                          ('type = "BatchedAsr"', 'type = "Asr"', "BatchedAsr")):
        p = tmp_path / "bad.toml"
        p.write_text(src.replace(old, new))
        with pytest.raises(config_toml.ConfigError, match=msg):
            config_toml.load_batched_asr(str(p))
    p = tmp_path / "f32.toml"
    p.write_text(src.replace("temperature = 0.0", 'temperature = 0.0\ndtype_override = "f32"'))
    cfg, _ = config_toml.load_batched_asr(str(p))
    assert cfg.kv_bf16 == 0
