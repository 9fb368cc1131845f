"""Committed golden vectors (tests/golden/tiny_asr.json, made by tools/make_golden.py from the CPU oracle).
CPU: the oracle still reproduces them (guards the checker against drift).  GPU: the HIP engine reproduces
them through the C ABI — a parity check that needs no oracle at run time."""
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tools", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _golden(mode=0):
    return json.load(open(os.path.join(ROOT, "tests", "golden", "tiny_asr_dot_mode1.json" if mode else "tiny_asr.json")))


def _norm(x):
    return json.loads(json.dumps(x))


@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_reproduces_golden(orc, tiny_weights, mode):
    got = _gen().run(orc.OracleAsr, lambda e, slot: e.mimi_reset_batch_idx(slot, side=0), dot_mode=mode)
    assert _norm(got) == _golden(mode)


def test_the_two_dot_modes_have_different_goldens():
    a, b = _golden(0), _golden(1)
    assert a["codes"] == b["codes"]        # Mimi (f32 weights) does not depend on the mode
    assert a["prs_bits"] != b["prs_bits"]  # the LM's floats do


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_engine_reproduces_golden(gpu, dsm, lib, tiny_weights, mode):
    got = _gen().run(dsm.AsrEngine, lambda e, slot: e.mimi_reset_batch_idx(slot), dot_mode=mode)
    want = _golden(mode)
    for key in ("codes", "text", "prs_bits", "msgs"):
        for s in range(want["steps"]):
            assert _norm(got[key][s]) == want[key][s], f"{key} differs from the golden vector at step {s}"
