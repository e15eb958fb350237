"""bwd(zstash) must equal bwd(recompute) bit for bit -- run against an alternative build to bisect (debug helper)"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
if len(sys.argv) > 1:
    B.LIB_PATH = os.path.abspath(sys.argv[1])
from so3x.so3_train import RotPredict
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
params = net.flat_params_nograd()
for n in (300, 4096, 1 << 17):
    x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
    t = torch.randint(0, 1000, (n,), device=dev)
    dout = torch.randn(n, 3, device=dev) / n
    out, zs = B.mlp_fwd_stash(params, x, t, 1000)
    g1 = B.mlp_bwd(params, x, t, dout, 1, 1000, zstash=zs)
    g0 = B.mlp_bwd(params, x, t, dout, 1, 1000)
    torch.cuda.synchronize()
    if n == 300:
        from oracle import oracle as O
        ref = O.mlp_bwd(params.cpu().numpy(), x.cpu().numpy(), t.cpu().numpy(), dout.cpu().numpy(), "f64")
        import numpy as np
        print("  vs oracle: recompute %.3e  stash %.3e  (scale %.3e)" % (np.abs(g0.cpu().numpy() - ref).max(), np.abs(g1.cpu().numpy() - ref).max(), np.abs(ref).max()))
    print(sys.argv[1:] or "in-tree", n, "equal" if torch.equal(g0, g1) else "DIFFERENT max %.3e (scale %.3e)" % (float((g0 - g1).abs().max()), float(g0.abs().max())))
