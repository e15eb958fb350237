"""one ProtNet training evaluation (exact-fp32 form, 256 complexes x (198 + 58)) for rocprofv3 --kernel-trace --stats"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-extensions_amd")]
import torch  # noqa: E402
from so3x import backend as B  # noqa: E402
from so3x.models import ProtNet  # noqa: E402
dev = torch.device("cuda:0")
n, lr, ll = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 198, 58
BIG = "--script-defaults" in sys.argv   # prot_train.py's argparse defaults: dim 1024, 8 heads, 12 layers, 8 convolutions
g = torch.Generator(device=dev).manual_seed(1)


def chains(L):
    res = torch.zeros(n * L, 21, device=dev)
    res[torch.arange(n * L, device=dev), torch.randint(0, 21, (n * L,), device=dev, generator=g)] = 1.0
    return (res, torch.randn(n * L, 3, device=dev, generator=g) * 8.0, B.quat_to_rmat(torch.randn(n * L, 4, device=dev, generator=g)).reshape(n * L, 9)), \
        torch.arange(0, n * L + 1, L, device=dev, dtype=torch.int64)


rec, roff = chains(lr)
lig, loff = chains(ll)
batch = B.ProtBatch(rec, lig, roff, loff, max(lr, ll), [(lr, ll)] * n)
t = torch.randint(0, 1000, (n,), device=dev, generator=g)
torch.manual_seed(0)
net = (ProtNet(dim=1024, heads=8, t_depth=12, c_depth=8) if BIG else ProtNet()).to(dev).train()
dout = torch.randn(n, 6, device=dev)
for _ in range(3):
    net.zero_grad(set_to_none=True)
    o = net(batch, t)
    (torch.cat((o.rot_g, o.shift_g), -1) * dout).sum().backward()
torch.cuda.synchronize()
