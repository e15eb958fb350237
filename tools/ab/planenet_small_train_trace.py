"""bf16 PlaneNet training evaluation at the reference's default shape (32 x 256) in a loop, for rocprofv3 --kernel-trace --stats"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x.models import PlaneNet
torch.manual_seed(0)
net = PlaneNet(precision="bf16", dropout=0.0).to("cuda:0").train()
x = torch.randn(32, 256, 3, device="cuda:0") * 0.5
t = torch.randint(0, 1000, (32,), device="cuda:0")
def step():
    net.zero_grad(set_to_none=True)
    net(x, t).square().sum().backward()
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print("wall us per step", (time.perf_counter() - t0) / 50 * 1e6)
