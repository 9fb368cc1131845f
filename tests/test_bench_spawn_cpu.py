"""`python bench.py --gpus N` must really start N ranks (the flag used to be parsed and ignored).
CPU rehearsal with gloo: the spawn path (parent never imports torch, children are started with torch.distributed.run as
child processes), the chunked load-time broadcast, barrier-bracketed timing with a MAX over ranks, ONE JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_that_many_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["broadcast_ok"] is True and out["steps"] == 3 and out["warmup"] == 1
    assert out["value"] >= 0.02  # MAX over ranks: rank 1 slept 20 ms


def _spawn(extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check", "--steps", "3", "--warmup", "1"] + extra,
                          capture_output=True, text=True, timeout=timeout, env=env)


def test_a_rank_that_dies_mid_run_fails_the_whole_bench_instead_of_hanging():
    """rank 1 raises between the two barriers of the timed region; rank 0 is parked in the second one.  The
    launcher has to take it down and `bench.py --gpus 2` has to return non-zero, without a JSON line, well inside the timeout."""
    r = _spawn(["--spawn-check-fail-rank", "1"], timeout=240)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout


def test_ranks_that_disagree_about_the_weight_arena_fail_the_bench():
    r = _spawn(["--spawn-check-arena-skew-rank", "1"], timeout=240)
    assert r.returncode != 0
    assert "disagree about the weight arena" in r.stdout + r.stderr


def test_rank0_line_carries_every_ranks_time():
    r = _spawn([])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert len(out["per_rank_ms"]) == 2 and out["per_rank_ms"][1] >= out["per_rank_ms"][0] * 0.5


def test_parent_does_not_import_torch_before_spawning():
    """The spawning parent must not initialise anything GPU-related: it may not even import torch."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main():")]
    top_level_imports = [ln for ln in head.splitlines() if ln.startswith(("import ", "from "))]
    assert not any("torch" in ln for ln in top_level_imports), top_level_imports
    main_body = src[src.index("def main():"):]
    assert main_body.index("spawn_ranks(args)") < main_body.index("import torch")
