"""Large-batch step time of the two-stream pipeline under the engine's tuning knobs (read from the environment at create):
one base engine loads the weights, every configuration is a fresh engine attached to the same weight arena.
    python experiments/large_batch_sweep.py [B ...]      -> one line per (B, knobs)"""
import itertools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, dsm_amd, bench
from dsm_amd import synth
cfg = dsm_amd.config_stt_1b_en_fr()
lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="stt-1b-en_fr")
base = dsm_amd.AsrEngine(cfg, 16, lm, mimi)
arena = base.weight_arena()
dev = torch.device("cuda", 0)
Bs = [int(x) for x in sys.argv[1:]] or [512, 2048]
KNOBS = [
    {},                                            # defaults: rolling window depth 4, 2 groups, graphs, staggered start from 256 slots per group
    {"DSM_STAGGER": "0"},
    {"DSM_LM_GROUPS": "4"},
    {"DSM_ATTN_LDS_PAD": "45000"},                 # three attention workgroups per CU instead of two
    {"DSM_ATTN_LDS_PAD": "0"},
    {"DSM_FUSE_FRONT": "0"},
    {"DSM_SMALLK_LOOP": "0"},
    {"DSM_ROLL": "0"},
]
for B in Bs:
    for kn in KNOBS:
        for k in ("DSM_ROLL", "DSM_LOOP_DEPTH", "DSM_LM_GROUPS", "DSM_GRAPHS", "DSM_ATTN_LDS_PAD", "DSM_GATE_OCC3", "DSM_STAGGER", "DSM_GEMM_LDS_PAD"):
            os.environ.pop(k, None)
        os.environ.update(kn)
        try:
            ms = bench.capacity_leg(dsm_amd, synth, cfg, B, arena, dev, 0, 12 if B <= 512 else 6, torch)
            print(json.dumps({"B": B, "knobs": kn, "ms_per_step": round(ms, 3), "rtf": round(80.0 / ms, 3)}), flush=True)
        except Exception as ex:
            print(json.dumps({"B": B, "knobs": kn, "error": str(ex)[:200]}), flush=True)
        torch.cuda.empty_cache()
base.close()
