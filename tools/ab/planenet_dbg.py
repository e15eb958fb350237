"""debug: bf16 PlaneNet, 65,536 tokens (256-tile GEMM) vs the same clouds in groups of four (128-tile GEMM)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x.models import PlaneNet
torch.manual_seed(0)
net = PlaneNet(precision="bf16", dropout=0.0).to("cuda:0").eval()
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gen = torch.Generator(device="cuda:0").manual_seed(5)
x = torch.randn(32, 2048, 3, device="cuda:0", generator=gen) * 0.5
t = torch.randint(0, 1000, (32,), device="cuda:0", generator=gen)
with torch.no_grad():
    big, ebig = net(x, t, want_encoding=True)
    for i in range(0, 32, 8):
        small, esmall = net(x[i:i + 4], t[i:i + 4], want_encoding=True)
        d = (ebig[i:i + 4] - esmall).abs()
        print(i, "out diff", float((big[i:i + 4] - small).abs().max()), "enc max", float(d.max()), "frac>0", float((d > 0).float().mean()),
              "frac>0.05", float((d > 0.05).float().mean()))
        if i == 0:
            bad = (d > 0.05).nonzero()
            print("bad count", bad.shape[0], "first", bad[:10].tolist())
            cols = torch.bincount(bad[:, 2], minlength=512)
            rows = torch.bincount(bad[:, 1] % 256, minlength=256)
            print("cols with bad", (cols > 0).nonzero().flatten()[:40].tolist())
            print("rows%256 with bad", (rows > 0).nonzero().flatten()[:40].tolist())
