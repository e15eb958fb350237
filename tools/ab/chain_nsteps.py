"""Launch duration of the reverse-chain kernel from the prepared state against the number of steps per launch (B = 2^20):
T(n) = a + b n separates the per-launch cost (a) from the steady-state step (b)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
n = 1 << 20
x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
ws, prec = proc._prepared(net)
_, trap_p = proc._tables()
out = torch.empty_like(x)
for ns in (1, 2, 3, 4, 8, 16, 100):
    def go():
        B.p_sample_prepared(ws, proc._sched, trap_p, x, 999, ns, seed=1, rng_offset=0, precision=prec, guide_p=proc._guide_p, out=out)
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"n_steps {ns:4d}: {us:8.1f} us per launch, {us / ns:7.2f} us per step")
