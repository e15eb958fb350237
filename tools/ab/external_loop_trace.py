"""The reference's one-call-per-step sampling loop (so3_test.py:28-31) for a kernel trace: 200 calls at B = 2^20.
rocprofv3 --kernel-trace --stats -d gpurun_out/ext -- python3 tools/ab/external_loop_trace.py"""
import sys, os, time
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
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
ts = [torch.full((1,), 999 - k, device=dev, dtype=torch.long) for k in range(200)]
for i in range(20):
    x = proc.p_sample(x, ts[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(200):
    x = proc.p_sample(x, torch.full((1,), 999 - k, device=dev, dtype=torch.long) if mode == "full" else ts[k])
torch.cuda.synchronize()
print(mode, "us per call", (time.perf_counter() - t0) / 200 * 1e6)
t0 = time.perf_counter()
for k in range(200):
    y = proc.p_sample(x, ts[k])   # host cost alone: no sync, then drain
el = time.perf_counter() - t0
torch.cuda.synchronize()
print("host issue us per call", el / 200 * 1e6)
