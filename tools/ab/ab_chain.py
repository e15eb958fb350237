import sys, os, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
GUIDE = True
if len(sys.argv) > 1:
    B.LIB_PATH = os.path.abspath(sys.argv[1])
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
_, trap_p = proc._tables()
params = net.flat_params_nograd()
n = 1 << 20
x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
def run():
    return B.p_sample_chain(params, proc._sched, trap_p, x, 600, 100, seed=1, precision=1, guide_p=(proc._guide_p if GUIDE else None))
for _ in range(2): run()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(sys.argv[1:] or "new", "ms", round(best, 4), "sample-steps/s %.4g" % (n * 100 / best * 1e3))
