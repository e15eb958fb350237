"""ProtNet training evaluation (forward + backward, exact-fp32 form, dropout 0.1) at the batch sizes prot_train.py actually uses
(default --batch 4): ms per evaluation by wall clock, i.e. with the host's launch overhead in it.
   python tools/ab/protnet_small_batch.py [batch ...] [--script-defaults]
(--script-defaults: the widths prot_train.py's argparse defaults give -- dim 1024, 8 heads, 12 encoder layers, 8 convolutions --
instead of the class defaults 64 / 4 / 4 / 3)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-extensions_amd")]
import torch  # noqa: E402
from so3x import backend as B  # noqa: E402
from so3x.models import ProtNet  # noqa: E402
dev = torch.device("cuda:0")
lr, ll = 198, 58
g = torch.Generator(device=dev).manual_seed(1)


def chains(n, L):
    res = torch.zeros(n * L, 21, device=dev)
    res[torch.arange(n * L, device=dev), torch.randint(0, 21, (n * L,), device=dev, generator=g)] = 1.0
    return (res, torch.randn(n * L, 3, device=dev, generator=g) * 8.0, B.quat_to_rmat(torch.randn(n * L, 4, device=dev, generator=g)).reshape(n * L, 9)), \
        torch.arange(0, n * L + 1, L, device=dev, dtype=torch.int64)


torch.manual_seed(0)
BIG = "--script-defaults" in sys.argv
net = (ProtNet(dim=1024, heads=8, t_depth=12, c_depth=8) if BIG else ProtNet()).to(dev).train()
print("parameters:", sum(p.numel() for p in net.parameters()))
for n in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [4, 16, 64]:
    rec, roff = chains(n, lr)
    lig, loff = chains(n, ll)
    batch = B.ProtBatch(rec, lig, roff, loff, max(lr, ll), [(lr, ll)] * n)
    t = torch.randint(0, 1000, (n,), device=dev, generator=g)
    dout = torch.randn(n, 6, device=dev)

    def step():
        net.zero_grad(set_to_none=True)
        o = net(batch, t)
        (torch.cat((o.rot_g, o.shift_g), -1) * dout).sum().backward()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    print(f"batch {n}: {best:.3f} ms per training evaluation (wall clock)")
