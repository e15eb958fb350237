"""A/B of the graph-replayed training step (2^19 samples, bf16) between builds of libso3x.so, interleaved in one process:
     python tools/ab/ab_trainlibs.py build/libso3x_a.so build/libso3x_b.so ... [--rounds 7] [--json out.json]
Each build gets its own network, optimizer and captured graph (the graph holds that build's kernels)."""
import sys, os, json, statistics, ctypes
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B, optim
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
from so3x.graphs import TrainStepGraph

args = sys.argv[1:]
rounds = int(args[args.index("--rounds") + 1]) if "--rounds" in args else 7
out = args[args.index("--json") + 1] if "--json" in args else None
libs = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] not in ("--rounds", "--json"))] or [B.LIB_PATH]
dev = "cuda:0"
n = 1 << 19
x0 = None
graphs = {}
for path in libs:
    B._lib, B.LIB_PATH = None, os.path.abspath(path)
    B.lib()
    torch.manual_seed(0)
    net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
    proc = SO3Diffusion(net, timesteps=1000).to(dev)
    opt = optim.Adam(net, lr=3e-4)
    if x0 is None:
        x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev))
    g = TrainStepGraph(proc, opt, x0.shape)
    for _ in range(20):
        g.replay()
    graphs[path] = (g, net, proc, opt)
torch.cuda.synchronize()
times = {p: [] for p in libs}
for r in range(rounds):
    for p in libs:
        g = graphs[p][0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        times[p].append(e0.elapsed_time(e1) / 50 * 1e3)
rows = []
for p in libs:
    med, mn = statistics.median(times[p]), min(times[p])
    rows.append({"build": os.path.basename(p), "us_per_step_median": round(med, 2), "us_min": round(mn, 2), "loss": float(graphs[p][0].loss)})
    print(f"{os.path.basename(p):28s} median {med:7.2f} us/step  min {mn:7.2f}  loss {float(graphs[p][0].loss):.4f}")
if out:
    json.dump({"what": "graph-replayed training step, 2^19 samples, bf16, interleaved rounds", "rounds": rounds, "rows": rows}, open(out, "w"), indent=1)
