import sys, os, cProfile, pstats, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
from so3x import optim as so3x_optim
which = sys.argv[1] if len(sys.argv) > 1 else "so3x"   # so3x | torch | torch-fused
opt = (so3x_optim.Adam(net, lr=3e-4) if which == "so3x" else
       torch.optim.Adam(net.parameters(), lr=3e-4, fused=(which == "torch-fused")))
n = 1 << 12   # tiny batch: host-bound
x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev))
def step():
    loss = proc(x0)
    opt.zero_grad()
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("host-bound step: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28); st.sort_stats("tottime").print_stats(22)
