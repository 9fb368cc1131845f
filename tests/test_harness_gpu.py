"""Builds and runs tests/harness/batched_asr_harness.cpp: the reference worker's two-thread call pattern
(srv/batched_asr.rs:314-522) against the C ABI from plain C++."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    out = os.path.join(ROOT, "tests", "harness", "batched_asr_harness")
    src = out + ".cpp"
    libdir = os.path.join(ROOT, "delayed-streams-modeling_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-o", out, "-L" + libdir, "-ldsm_mi355x", "-lpthread",
                           "-Wl,-rpath," + libdir])
    return out


def test_harness_compiles_against_the_header(lib):
    """CPU: the harness only needs include/dsm.h and the shared library to link."""
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_two_thread_pipeline_matches_sequential(gpu, dsm, lib, tiny_weights):
    exe = _build()
    r = subprocess.run([exe, tiny_weights[0], tiny_weights[1], "5", "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "harness ok" in r.stdout
