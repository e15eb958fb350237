"""measure: bf16 one-kernel training step vs the fp32 path of the same stack on identical draws (200 Adam steps, B = 2^15)"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
from so3x import optim as so3x_optim
dev = "cuda:0"
torch.manual_seed(0)
n16 = RotPredict(out_type="skewvec", precision="bf16").to(dev)
n32 = RotPredict(out_type="skewvec", precision="fp32").to(dev)
n32.load_state_dict(n16.state_dict())
theta0 = n16.flat_data().clone()
T, Bn, steps = 1000, 1 << 15, int(sys.argv[1]) if len(sys.argv) > 1 else 200
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-4
p16, p32 = SO3Diffusion(n16, timesteps=T).to(dev), SO3Diffusion(n32, timesteps=T).to(dev)
o16, o32 = so3x_optim.Adam(n16, lr=lr), so3x_optim.Adam(n32, lr=lr)
z90 = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
rot = torch.stack((z90, z90.T), 0).to(dev)
g = torch.Generator(device=dev).manual_seed(7)
l16, l32 = [], []
for k in range(steps):
    x0 = rot[torch.randint(0, 2, (Bn,), device=dev, generator=g)]
    t = torch.randint(0, T, (Bn,), device=dev, generator=g)
    ax, un = torch.randn(Bn, 3, device=dev, generator=g), torch.rand(Bn, device=dev, generator=g)
    for proc, opt, acc in ((p16, o16, l16), (p32, o32, l32)):
        loss = proc.p_losses(x0, t, axes=ax, unif=un)
        opt.zero_grad()
        loss.backward()
        opt.step()
        acc.append(loss.detach())
l16, l32 = torch.stack(l16).cpu(), torch.stack(l32).cpu()
rel = ((l16 - l32).abs() / l32)
print("loss first/last fp32", float(l32[0]), float(l32[-1]), "bf16", float(l16[0]), float(l16[-1]))
print("max rel loss diff", float(rel.max()), "mean", float(rel.mean()), "last 20 mean", float(rel[-20:].mean()))
d = (n16.flat_data() - n32.flat_data()).norm()
trav = (n32.flat_data() - theta0).norm()
print("param drift", float(d), "travelled", float(trav), "ratio", float(d / trav), "rel to |theta|", float(d / n32.flat_data().norm()))
