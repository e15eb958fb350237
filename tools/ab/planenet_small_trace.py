"""bf16 PlaneNet inference at the reference's default shape (32 clouds x 256 points) in a loop, for rocprofv3 --kernel-trace --stats"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x.models import PlaneNet
torch.manual_seed(0)
net = PlaneNet(precision="bf16", dropout=0.0).to("cuda:0").eval()
x = torch.randn(32, 256, 3, device="cuda:0") * 0.5
t = torch.randint(0, 1000, (32,), device="cuda:0")
with torch.no_grad():
    for _ in range(20):
        net(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        net(x, t)
    torch.cuda.synchronize()
    print("wall us per forward", (time.perf_counter() - t0) / 100 * 1e6)
