"""The graph-replayed training step at BASELINE config 4's shard size, for `rocprofv3 --kernel-trace --stats` and for a
quick wall-clock number:   python tools/ab/train_prof.py [log2_batch=19] [replays=200] [optimizer=so3x|torch]"""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B, optim
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
from so3x.graphs import TrainStepGraph
dev = "cuda:0"
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
which = sys.argv[3] if len(sys.argv) > 3 else "so3x"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
opt = optim.Adam(net, lr=3e-4) if which == "so3x" else torch.optim.Adam(net.parameters(), lr=3e-4, fused=True, capturable=True)
n = 1 << lg
x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev))
def eager():
    loss = proc(x0)
    opt.zero_grad()
    loss.backward()
    opt.step()
for _ in range(5): eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): eager()
torch.cuda.synchronize()
print("eager  n=2^%d: %.1f us/step" % (lg, (time.perf_counter() - t0) / 50 * 1e6))
g = TrainStepGraph(proc, opt, x0.shape)
for _ in range(20): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): g.replay()
torch.cuda.synchronize()
print("graph  n=2^%d (%s Adam): %.1f us/step, loss %.4f" % (lg, which, (time.perf_counter() - t0) / reps * 1e6, float(g.loss)))
