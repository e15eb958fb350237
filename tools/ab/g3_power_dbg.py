"""measure: G3's statistics for the shipped bf16 chain and for deliberately wrong samplers (sigma_t scaled, network output scaled)"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import oracle as O
from so3x import backend as B, rng
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
dev = "cuda:0"
g = np.load(os.path.join(ROOT, "tests", "golden", "chain_samples_trained.npz"))
ref = g["x_final"]; m = len(ref)
net = RotPredict(out_type="skewvec", precision="bf16")
net.load_state_dict({f"net.{l}.{k}": torch.from_numpy(g[f"net_{l}_{k}"]) for l in (0, 2, 4, 6, 8) for k in ("weight", "bias")})
net = net.to(dev)
thr = O.ker_2samp_threshold(m)
noise = max(O.MMD(ref[: m // 2], ref[m // 2:]), 1e-3)
z90 = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
def stats(X):
    d0 = O.rmat_dist(X, np.broadcast_to(z90, X.shape).copy(), "f64"); d1 = O.rmat_dist(X, np.broadcast_to(z90.T, X.shape).copy(), "f64")
    return np.median(np.minimum(d0, d1)), np.mean(d0 < d1)
print("thr", thr, "5x noise", 5 * noise, "ref stats", stats(ref.astype(np.float64)))
def run(tag, sig=1.0, vscale=1.0, seed=2024):
    n2 = copy.deepcopy(net)
    with torch.no_grad():
        n2.net[8].weight.mul_(vscale); n2.net[8].bias.mul_(vscale)
    proc = SO3Diffusion(n2, timesteps=1000).to(dev)
    if sig != 1.0:
        proc._tables()
        proc._sched[12] *= sig
        proc._trap_p = B.igso3_build_tables(proc._sched[12])
        proc._guide_p = B.igso3_build_guide(proc._trap_p)
    rng.manual_seed(seed)
    x = proc.p_sample_loop((m,)).cpu().numpy().astype(np.float64)
    print(tag, "MMD", O.MMD(x, ref), "stats", stats(x))
run("shipped")
run("shipped seed 7", seed=7)
for s in (1.05, 1.1, 1.15, 1.3): run(f"sigma x{s}", sig=s)
for v in (0.9, 0.8, 1.1): run(f"v x{v}", vscale=v)
