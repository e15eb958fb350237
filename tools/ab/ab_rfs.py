import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
if len(sys.argv) > 1:
    B.LIB_PATH = os.path.abspath(sys.argv[1])
dev = "cuda:0"
torch.manual_seed(0)
pw = torch.randn(B.N_PARAMS_RESNET, device=dev) * 0.06
n = 1 << 19
x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
t = torch.randint(0, 1000, (n,), device=dev)
def run():
    return B.resnet_fwd_stash(pw, x, t, 1000, 1)
for _ in range(3): run()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(sys.argv[1:] or "base", "resnet_fwd_stash ms", round(best, 4))
