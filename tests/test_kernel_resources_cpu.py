"""Register / LDS budget of the hot kernels, from the device assembly (no GPU): the r04 lesson that a static LDS array which bounds
the occupancy makes the compiler pad a kernel's VGPR allocation (136 allocated for 88 used in gemm_bx3u_kernel and
seanet_front_kernel) — registers the other streams' workgroups then cannot use beside it — is pinned here: no kernel of the library
may allocate 16 or more VGPRs beyond what it uses, and the kernels of the B = 64 step must not spill."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc (device assembly)")
def test_no_padded_register_allocation_and_no_spills_on_the_hot_path():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [l for l in out.stdout.splitlines()[1:] if l.strip()]
    assert len(rows) > 100, "kernel table looks empty"
    padded = [l for l in rows if l.endswith("PADDED")]
    assert not padded, "VGPR allocation padded beyond use (static LDS bounding the occupancy?):\n" + "\n".join(padded)
    hot = ("gemm_bx3u_kernel<", "attn_kernel<bf16, 128, 1, 8>", "attn_kernel<bf16, 64, 1, 4>", "attn_small_kernel<", "gemm_wk_kernel<",
           "gemm_reduce_rows_kernel<", "gemm_reduce_kernel<", "seanet_front_kernel<", "rvq_select_kernel", "dep_argmax_kernel", "row_norm_kernel")
    spills = [l for l in rows if l.endswith("SPILLS") and any(l.startswith(h) for h in hot)]
    assert not spills, "a hot-path kernel spills:\n" + "\n".join(spills)
    # the budget DESIGN §9 item 0 describes: the 32-row GEMMs and the head_dim-64 attention stay within 104 / 80 VGPRs
    alloc = {l[:64].strip(): int(l[64:].split()[1]) for l in rows}
    assert alloc["gemm_bx3u_kernel<bf16, 2, 1, 1, 8, 0, 4>"] <= 80 and alloc["gemm_bx3u_kernel<bf16, 2, 2, 2, 8, 0, 4>"] <= 104
    assert alloc["attn_kernel<bf16, 64, 1, 4>"] <= 80 and alloc["attn_kernel<bf16, 128, 1, 8>"] <= 128
