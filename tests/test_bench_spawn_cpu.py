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


def test_parent_does_not_import_torch_before_spawning():
    """The spawning parent must not initialise anything GPU-related: it may not even import torch."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main():")]
    top_level_imports = [ln for ln in head.splitlines() if ln.startswith(("import ", "from "))]
    assert not any("torch" in ln for ln in top_level_imports), top_level_imports
    main_body = src[src.index("def main():"):]
    assert main_body.index("spawn_ranks(args)") < main_body.index("import torch")
