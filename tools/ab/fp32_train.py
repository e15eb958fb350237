"""a few fp32 training steps of the 65-wide network at 2^19 samples (run under rocprofv3 --kernel-trace --stats)"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
dev = "cuda:0"
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
net = RotPredict(out_type="skewvec", precision=prec).to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=3e-4, fused=True)
x0 = B.quat_to_rmat(torch.randn(1 << 19, 4, device=dev))
for _ in range(6):
    loss = proc(x0)
    opt.zero_grad()
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("loss", float(loss.detach()))
