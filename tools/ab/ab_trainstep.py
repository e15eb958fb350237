"""One number for a build of the library: the graph-replayed 65-wide training step at 2^19 samples (bf16), us per step.
     LD_PRELOAD=build/libso3x_<variant>.so python tools/ab/ab_trainstep.py [rounds=5] [tag]
(preloading a variant build makes the operator library bind the so3x_* entry points to it; no LD_PRELOAD = the product library).
Run the variants back to back in ONE gpurun call (same box); compare medians."""
import sys, os, json, statistics
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B, optim
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
from so3x.graphs import TrainStepGraph

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tag = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("LD_PRELOAD", "product")
dev = "cuda:0"
n = 1 << 19
torch.manual_seed(0)
x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev))
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
opt = optim.Adam(net, lr=3e-4)
g = TrainStepGraph(proc, opt, x0.shape, pipeline=False)
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
ts = []
for r in range(rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 200 * 1e3)
grad = net.flat_grad()
print(json.dumps({"variant": tag, "us_per_step_median": round(statistics.median(ts), 2), "us_min": round(min(ts), 2), "loss": float(g.loss),
                  "param_checksum": float(net.flat_data().double().sum()), "grad_abs_sum": float(grad.double().abs().sum()) if grad is not None else None}))
