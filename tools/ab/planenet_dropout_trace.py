"""bf16 PlaneNet training evaluations at 32 x 2048 with a given dropout probability, for a kernel trace (what dropout costs, kernel by kernel):
   rocprofv3 --kernel-trace --output-format csv -d out -o p -- python3 tools/ab/planenet_dropout_trace.py 0.1   (and 0.0)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x.models import PlaneNet
torch.manual_seed(0)
net = PlaneNet(precision="bf16", dropout=float(sys.argv[1])).to("cuda:0").train()
x = torch.randn(32, 2048, 3, device="cuda:0") * 0.5
t = torch.randint(0, 1000, (32,), device="cuda:0")
for _ in range(6):
    net.zero_grad(set_to_none=True)
    net(x, t).square().sum().backward()
torch.cuda.synchronize()
