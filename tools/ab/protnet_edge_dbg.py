"""bf16 vs exact-fp32 ProtNet forward on the same weights: where the difference sits (encoder output per chain / head input / output)"""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/diffusion-extensions_amd", ROOT + "/tests"]
import torch, warnings; warnings.filterwarnings("ignore")
from test_protnet import synthetic_complexes, to_dev, protnet_perturb
from so3x.models import ProtNet
from so3x import backend as B
DEV = "cuda:0"
torch.manual_seed(5)
net = ProtNet(precision="bf16", dropout=0.0).eval()
protnet_perturb(net, 9)
net = net.to(DEV)
lengths = [(198, 58), (40, 256), (129, 77), (64, 65)]
data = B.ProtBatch.from_pairs(to_dev(synthetic_complexes(lengths, 77)))
t = torch.randint(0, 1000, (len(lengths),), device=DEV)
for T in (1, 2, 4):
    cfg = (64, 4, T, 3)
    import so3x.models as M
    torch.manual_seed(5)
    n2 = ProtNet(t_depth=T, precision="bf16", dropout=0.0).eval(); protnet_perturb(n2, 9); n2 = n2.to(DEV)
    with torch.no_grad():
        o16, _, p16, e16 = B.protnet_fwd(n2.flat_params_nograd(), data, t, *n2.cfg, precision=B.PREC_BF16, want_pool=True, want_encoding=True)
        o32, _, p32, e32 = B.protnet_fwd(n2.flat_params_nograd(), data, t, *n2.cfg, precision=B.PREC_F32, want_pool=True, want_encoding=True)
    Bn = len(lengths)
    worst = []
    for i, (lr, ll) in enumerate(lengths):
        for s_, L in ((i, lr), (Bn + i, ll)):
            a, b = e16[s_, :L], e32[s_, :L]
            worst.append((float((a - b).abs().max()), float((a - b).abs().mean()), L))
    print("t_depth", T, "enc max/mean abs err per chain", [(round(w[0], 3), round(w[1], 4), w[2]) for w in worst], "enc scale", float(e32.abs().max()))
    d = 64
    print("   out rel", float((o16 - o32).abs().max() / o32.abs().max()), "rec pool rel", float((p16[:, d:2*d] - p32[:, d:2*d]).abs().max() / p32[:, d:2*d].abs().max()))
