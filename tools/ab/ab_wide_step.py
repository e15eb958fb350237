"""the wide network's whole training step as bench.py's wide_net leg runs it (one captured hipGraph per step, so3x.optim.Adam), 2^19
samples; to compare builds copy one over diffusion-extensions_amd/libso3x.so on the GPU box between two runs (the argument is a label)"""
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B, optim
from so3x.so3_lock_train import RotPredict
from so3x.diffusion import SO3Diffusion
from so3x.graphs import TrainStepGraph
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
n = 1 << 19
x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev))
tg = TrainStepGraph(proc, optim.Adam(net, lr=3e-4), x0.shape)
for _ in range(3):
    tg.replay()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        tg.replay()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print(sys.argv[1:] or "in-tree", "wide-net training step ms %.4f" % best, "loss", float(tg.loss))
