"""A/B of the training step's forward half inside the replayed graph: the two-launch form (default: noising kernel, then forward
with the MSE epilogue) against the fused noising + forward kernel (SO3X_AB_TRAINFWD=fused), interleaved rounds, 2^19 samples, bf16.
     LD_PRELOAD=diffusion-extensions_amd/libso3x_ab.so python tools/ab/ab_trainfwd.py [rounds=7] [out.json]
The switch exists only in the A/B build (libso3x_ab.so, -DSO3X_AB_BUILD); preloading it makes the operator library bind the
so3x_* entry points to it instead of the product library, which reads no environment."""
import sys, os, json, statistics
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B, optim
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
from so3x.graphs import TrainStepGraph

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = "cuda:0"
n = 1 << 19
x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev))
graphs = {}
for name, env in (("split", None), ("fused", "fused")):
    if env:
        os.environ["SO3X_AB_TRAINFWD"] = env      # read by the launcher at capture time
    else:
        os.environ.pop("SO3X_AB_TRAINFWD", None)
    torch.manual_seed(0)
    net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
    proc = SO3Diffusion(net, timesteps=1000).to(dev)
    opt = optim.Adam(net, lr=3e-4)
    g = TrainStepGraph(proc, opt, x0.shape)
    for _ in range(20):
        g.replay()
    graphs[name] = (g, net)
os.environ.pop("SO3X_AB_TRAINFWD", None)
torch.cuda.synchronize()
times = {k: [] for k in graphs}
for r in range(rounds):
    for name, (g, _) in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 50 * 1e3)
rows = []
for name in graphs:
    med, mn = statistics.median(times[name]), min(times[name])
    rows.append({"variant": name, "us_per_step_median": round(med, 2), "us_min": round(mn, 2), "loss": float(graphs[name][0].loss)})
    print(f"{name:8s} median {med:7.2f} us/step  min {mn:7.2f}  loss {float(graphs[name][0].loss):.4f}")
same = torch.equal(graphs["fused"][1].flat_data(), graphs["split"][1].flat_data())
print("parameters after the same number of steps identical:", same)
if len(sys.argv) > 2:
    json.dump({"what": "graph-replayed training step, 2^19 samples, bf16: fused noising + forward kernel vs two launches", "rounds": rounds,
               "rows": rows, "bit_identical_parameters": same}, open(sys.argv[2], "w"), indent=1)
