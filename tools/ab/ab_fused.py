#!/usr/bin/env python3
"""A/B of the training step's device work at BASELINE config 4's shard size (2^19 rotations): the one-kernel step
(so3x_train_fused + so3x_train_bwd_reduce_adam) against the staged step (so3x_train_noise / _net / _bwd_partial +
so3x_train_bwd_reduce_adam), both as replayed hipGraphs, interleaved rounds, median.  usage: ab_fused.py [log2 n] [out.json]"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-extensions_amd")]
import torch  # noqa: E402
from so3x import backend as B  # noqa: E402
from so3x.diffusion import SO3Diffusion  # noqa: E402
from so3x.so3_train import RotPredict  # noqa: E402

DEV = "cuda:0"


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    n = 1 << lg
    torch.manual_seed(0)
    net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    proc = SO3Diffusion(net, timesteps=1000).to(DEV)
    trap_q, _ = proc._tables()
    x0 = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
    params = net.flat_data().clone()
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    step = torch.zeros(2, device=DEV)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    buf = B.TrainBuffers(n, 1000, DEV)

    def tail():
        B.train_bwd_reduce_adam(buf, params, m, v, step, 3e-4, 0.9, 0.999, 1e-8)

    def fused():
        B.train_fused(buf, params, proc._sched, trap_q, x0, None, seed=1, rng_counter=ctr, guide_q=proc._guide_q)
        tail()

    def staged():
        B.train_noise(buf, proc._sched, trap_q, x0, None, seed=1, rng_counter=ctr, guide_q=proc._guide_q)
        B.train_net(buf, params, rng_counter=ctr)
        B.train_bwd_partial(buf)
        tail()

    graphs = {}
    for name, fn in (("fused", fused), ("staged", staged)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        graphs[name] = g
    times = {k: [] for k in graphs}
    for _ in range(9):
        for name, g in graphs.items():
            for _ in range(5):
                g.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(40):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            times[name].append(a.elapsed_time(b) / 40 * 1e3)
    res = {"n": n, "us_per_step": {k: round(statistics.median(v), 2) for k, v in times.items()},
           "all": {k: [round(x, 1) for x in v] for k, v in times.items()}, "loss": float(buf.loss[0])}
    print(json.dumps(res))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
