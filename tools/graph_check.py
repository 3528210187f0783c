"""hipGraph replay of the inference step vs eager launches: equality and ms/step.  python tools/graph_check.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
import bench
from tgsr_amd.miscc.config import cfg, cfg_reset
from tgsr_amd.trainer import SRPipeline
from tgsr_amd.synthetic import synthetic_batch
cfg_reset(); cfg.GAN.GF_DIM = 32; cfg.TEXT.EMBEDDING_DIM = 256; cfg.TREE.BRANCH_NUM = 4; cfg.TREE.BASE_SIZE = 32
dev = torch.device("cuda:0")
pipe = SRPipeline(41, device=dev, low="lr", overlap=True, branch_num=4)
w = bench.load_weights(); pipe.load_state_dicts(w["E."], w["GL."], w["GH."])
cap, lens, LR, LRb = synthetic_batch(BATCH, seed=100)
cap, LR, LRb = cap.to(dev), LR.to(dev), LRb.to(dev); lens = lens.tolist()
torch.manual_seed(0)
ref = pipe(cap, lens, LR, LRb); ref_fine = [f.clone() for f in ref["fine"]]
out = pipe.capture(cap, lens, LR, LRb)
out = pipe.replay(cap, LR, LRb); torch.cuda.synchronize()
print("max diff graph vs eager:", max(float((a - b).abs().max()) for a, b in zip(out["fine"], ref_fine)))
cap2, lens2, LR2, LRb2 = synthetic_batch(BATCH, seed=101)
o2 = pipe.replay(cap.clone(), LR2.to(dev), LRb2.to(dev)); torch.cuda.synchronize()
e2 = pipe(cap, lens, LR2.to(dev), LRb2.to(dev))
print("new inputs: max diff", max(float((a - b).abs().max()) for a, b in zip(o2["fine"], e2["fine"])))
for name, fn in (("eager", lambda: pipe(cap, lens, LR, LRb)), ("graph", lambda: pipe.replay(cap, LR, LRb))):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(name, "%.3f ms/step  %.0f img/s" % (dt * 1e3, BATCH / dt))
