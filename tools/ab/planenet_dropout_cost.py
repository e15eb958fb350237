"""One PlaneNet bf16 training evaluation (forward with stash + backward) at 32 x 2048 with and without the reference's dropout 0.1."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x.models import PlaneNet
torch.manual_seed(0)
x = torch.randn(32, 2048, 3, device="cuda:0") * 0.5
t = torch.randint(0, 1000, (32,), device="cuda:0")
for p in (0.0, 0.1):
    net = PlaneNet(precision="bf16", dropout=p).to("cuda:0").train()
    def step():
        net.zero_grad(set_to_none=True)
        net(x, t).square().sum().backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"dropout {p}: {e0.elapsed_time(e1) / 10:.3f} ms per training evaluation")
