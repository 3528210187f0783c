import faulthandler, os, sys
faulthandler.enable()
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgsr_amd.miscc.config import cfg
cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM = 32, 256
from tgsr_amd.synthetic import synthetic_batch
from tgsr_amd.trainer import SRPipeline, GraphedStep
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
pipe = SRPipeline(41, device="cuda", dtype=dt)
cap, lens, LR, LRb = synthetic_batch(16, seed=100)
cap, LR, LRb, lens = cap.cuda(), LR.cuda(), LRb.cuda(), lens.tolist()
ref = pipe(cap, lens, LR, LRb)["fine"][2].clone()
print("eager ok", flush=True)
g = GraphedStep(pipe, cap, lens, LR, LRb, lanes=2)
print("captured", flush=True)
out = g.replay()
torch.cuda.synchronize()
for k in range(2):
    print("lane", k, float((out[k]["fine"][2] - ref).abs().max()), flush=True)
