"""debug: bf16 PlaneNet backward vs the exact-fp32 form on the same weights and inputs (per-parameter relative errors)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x.models import PlaneNet
Bn, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 256)
torch.manual_seed(21)
nets = {}
for prec in ("fp32", "bf16"):
    torch.manual_seed(21)
    nets[prec] = PlaneNet(precision=prec, dropout=0.0).to("cuda:0").train()
gen = torch.Generator(device="cuda:0").manual_seed(5)
x = torch.randn(Bn, P, 3, device="cuda:0", generator=gen) * 0.5
t = torch.randint(0, 1000, (Bn,), device="cuda:0", generator=gen)
dout = torch.randn(Bn, 3, device="cuda:0", generator=gen)
grads = {}
for prec, net in nets.items():
    out = net(x, t)
    (out * dout).sum().backward()
    torch.cuda.synchronize()
    grads[prec] = {k: p.grad.clone() for k, p in net.named_parameters()}
    print(prec, "out", out[0].tolist())
worst = 0
for k in grads["fp32"]:
    a, b = grads["fp32"][k], grads["bf16"][k]
    rel = float((a - b).norm() / (a.norm() + 1e-30))
    mx = float((a - b).abs().max() / (a.abs().max() + 1e-30))
    worst = max(worst, rel)
    print(f"{k:50s} |g| {float(a.norm()):.3e}  rel-norm-err {rel:.3e}  max-err/max {mx:.3e}  finite {bool(torch.isfinite(b).all())}")
print("worst rel", worst)
net = nets["bf16"]
for _ in range(2):
    net.zero_grad(set_to_none=True); (net(x, t) * dout).sum().backward()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5):
    net.zero_grad(set_to_none=True); (net(x, t) * dout).sum().backward()
torch.cuda.synchronize(); print("bf16 fwd+bwd ms", (time.time() - t0) / 5 * 1e3)
