"""Is the host the limit?  Enqueue time per step (host returns from the last launch of a step) vs total time per step, at
stt-1b B = 64, for the eager path (DSM_GRAPHS=0) and the hipGraph replay (default).  Run once per mode:
    DSM_GRAPHS=0 python experiments/host_enqueue_rate.py ; python experiments/host_enqueue_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dsm_amd
from dsm_amd import synth
cfg = dsm_amd.config_stt_1b_en_fr(); B = int(os.environ.get("B", "64"))
lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="stt-1b-en_fr")
eng = dsm_amd.AsrEngine(cfg, B, lm, mimi)
dev = torch.device("cuda", 0)
pcm = torch.from_numpy(synth.synth_pcm(B, 4)).to(dev); mask = torch.ones(B, dtype=torch.uint8, device=dev)
text = torch.zeros(B, dtype=torch.int32, device=dev); prs = torch.zeros(4 * B, device=dev); codes = torch.zeros(B * 32, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
eng.debug_set_positions(3000, 1000)
def step(i):
    eng.encode_step_dev(pcm[i % 4].data_ptr(), mask.data_ptr(), codes.data_ptr())
    eng.step_tokens_dev(None, mask.data_ptr(), text.data_ptr(), prs.data_ptr())
for i in range(10): step(i)
torch.cuda.synchronize()
for n in (20, 100):
    t0 = time.perf_counter()
    for i in range(n): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    m = eng.metrics()
    print(f"DSM_GRAPHS={os.environ.get('DSM_GRAPHS', '1')} B={B} n={n}: host enqueue {1e3*(t1-t0)/n:.3f} ms/step, total {1e3*(t2-t0)/n:.3f} ms/step "
          f"(graph launches {m.graph_launches}, eager bodies {m.eager_bodies})")
