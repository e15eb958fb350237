import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x.models import PlaneNet
torch.manual_seed(0)
net = PlaneNet(precision="bf16", dropout=0.0).to("cuda:0").eval()
x = torch.randn(32, 2048, 3, device="cuda:0") * 0.5
t = torch.randint(0, 1000, (32,), device="cuda:0")
with torch.no_grad():
    for _ in range(30):
        net(x, t)
torch.cuda.synchronize()
