#!/usr/bin/env python3
"""PlaneNet forward (and backward) timing on one MI355X: the reference's default batch (32 clouds x 256 points,
aircraft_rotate.py:17-30) and 32 x 2048, HIP events around back-to-back calls of the operator.  Prints one JSON line per shape.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch

from so3x import backend as B
from so3x.models import PlaneNet


def planenet_flop(Bn, P, dim=512, heads=4, layers=4, ffn=2048):
    """multiply-adds x 2 of the forward: post_scale, per layer in_proj + QK^T + PV + out_proj + the two feed-forward products"""
    n = Bn * P
    per_tok = 2 * (dim // 2) ** 2 + layers * (2 * dim * 3 * dim + 4 * P * dim + 2 * dim * dim + 4 * dim * ffn)
    return n * per_tok + Bn * (2 * dim * dim + 6 * dim)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    precisions = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16"]
    for prec in precisions:
        net = PlaneNet(precision=prec, dropout=0.0).to(dev).eval()
        for Bn, P in ((32, 256), (32, 2048)):
            x = torch.randn(Bn, P, 3, device=dev) * 0.5
            t = torch.randint(0, 1000, (Bn,), device=dev)
            with torch.no_grad():
                for _ in range(3):
                    net(x, t)
                ms = min(timed(lambda: net(x, t), 10 if prec == "bf16" else 2) for _ in range(3))
            fl = planenet_flop(Bn, P)
            rec = {"precision": prec, "clouds": Bn, "points": P, "forward_ms": ms, "forward_TFLOPs": fl / ms / 1e9,
                   "frac_of_bf16_peak": fl / ms / 1e9 / 2500.0, "flop": fl}
            # one training evaluation: forward with the stash + backward (dX and dW of every product: 2 x the forward's flops;
            # attention recomputes its probabilities: 2.5 x)
            net.train()
            dout = torch.randn(Bn, 3, device=dev)

            def step():
                net.zero_grad(set_to_none=True)
                (net(x, t) * dout).sum().backward()
            for _ in range(2):
                step()
            ms_t = min(timed(step, 5 if prec == "bf16" else 1) for _ in range(2))
            attn = Bn * P * 4 * 4 * P * 512
            fl_t = 3 * (fl - attn) + 3.5 * attn
            rec.update({"train_eval_ms": ms_t, "train_eval_TFLOPs": fl_t / ms_t / 1e9, "train_eval_frac_of_bf16_peak": fl_t / ms_t / 1e9 / 2500.0,
                        "train_eval_flop": fl_t})
            net.eval()
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
