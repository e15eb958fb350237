"""measure: PlaneNet bf16 forward at 32 x 2048 as one call vs as sub-batches of clouds (do the intermediates stay in the Infinity Cache?)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x.models import PlaneNet
torch.manual_seed(0)
net = PlaneNet(precision="bf16", dropout=0.0).to("cuda:0").eval()
x = torch.randn(32, 2048, 3, device="cuda:0") * 0.5
t = torch.randint(0, 1000, (32,), device="cuda:0")
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
with torch.no_grad():
    for _ in range(3): net(x, t)
    for chunk in (32, 16, 8, 4):
        def run():
            for i in range(0, 32, chunk):
                net(x[i:i + chunk], t[i:i + chunk])
        print(chunk, "clouds per call:", min(timed(run) for _ in range(3)), "ms for all 32")
