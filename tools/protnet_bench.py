"""ProtNet forward at BASELINE config 5's shape (4096 complexes x 256 residues = a 198-residue receptor + a 58-residue ligand, the
BPTI docking set's typical split): time per forward by HIP events, algorithmic flops / time / 2.5 PF.  usage: python tools/protnet_bench.py [B] [precision]"""
import os
import sys
import time
from collections import namedtuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-extensions_amd")]
import torch  # noqa: E402
from so3x import backend as B  # noqa: E402
from so3x.models import ProtNet  # noqa: E402

ProtData = namedtuple("ProtData", ["residues", "positions", "angles"])


def flops(lengths, dim=64, heads=4, t_depth=4, c_depth=3, ffn=2048):
    """multiply-adds x 2 of every Linear / Conv1d and of attention's two products"""
    tot = 0
    for L in lengths:
        conv = 21 * dim * 3 + (c_depth - 2) * dim * dim * 3 + dim * (dim - dim // 2 - dim // 4) * 3
        siren = 3 * (dim // 2) + (dim // 2) ** 2 + 9 * (dim // 4) + (dim // 4) ** 2
        layer = 3 * dim * dim + dim * dim + 2 * dim * ffn + 2 * L * dim
        tot += 2 * L * (conv + siren + t_depth * layer + dim + dim)
    return tot


def main():
    Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = ProtNet(precision=prec).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1)
    lr, ll = 198, 58

    def chains(n, L):
        res = torch.zeros(n * L, 21, device=dev)
        res[torch.arange(n * L, device=dev), torch.randint(0, 21, (n * L,), device=dev, generator=g)] = 1.0
        pos = torch.randn(n * L, 3, device=dev, generator=g) * 8.0
        ang = B.quat_to_rmat(torch.randn(n * L, 4, device=dev, generator=g)).reshape(n * L, 9)
        return (res, pos, ang), torch.arange(0, n * L + 1, L, device=dev, dtype=torch.int64)
    rec, roff = chains(Bn, lr)
    lig, loff = chains(Bn, ll)
    batch = B.ProtBatch(rec, lig, roff, loff, max(lr, ll), [(lr, ll)] * Bn)
    t = torch.randint(0, 1000, (Bn,), device=dev, generator=g)
    fl = flops([lr] * Bn + [ll] * Bn)
    with torch.no_grad():
        for _ in range(3):
            out = net(batch, t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10 if prec == "bf16" else 2
        e0.record()
        for _ in range(reps):
            out = net(batch, t)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print({"complexes": Bn, "residues": Bn * (lr + ll), "precision": prec, "ms": ms, "TFLOPs": fl / ms / 1e9, "frac_of_2.5PF": fl / ms / 1e9 / 2500.0,
           "finite": bool(torch.isfinite(out.rot_g).all())})


if __name__ == "__main__":
    main()
