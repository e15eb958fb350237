import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
if len(sys.argv) > 1:
    B.LIB_PATH = os.path.abspath(sys.argv[1])
from so3x.so3_train import RotPredict
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
params = net.flat_params_nograd()
n = 1 << 19
x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
t = torch.randint(0, 1000, (n,), device=dev)
dout = torch.randn(n, 3, device=dev) / n
out, zs = B.mlp_fwd_stash(params, x, t, 1000)
def run():
    return B.mlp_bwd(params, x, t, dout, 1, 1000, zstash=zs)
for _ in range(3): run()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
bwd_ms = best
best = 1e9
for _ in range(3): B.mlp_fwd_stash(params, x, t, 1000)
torch.cuda.synchronize()
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); B.mlp_fwd_stash(params, x, t, 1000); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(sys.argv[1:] or "in-tree", "mlp_bwd ms", round(bwd_ms, 4), "mlp_fwd_stash ms", round(best, 4))
