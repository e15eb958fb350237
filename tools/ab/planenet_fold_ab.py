"""bf16 PlaneNet inference at 32 x 2048: the LayerNorm-folded path (no stash) against the plain kernel sequence (what a forward with
a stash runs), same process, interleaved."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x import backend as B
from so3x.models import PlaneNet
torch.manual_seed(0)
net = PlaneNet(precision="bf16", dropout=0.0).to("cuda:0").eval()
x = torch.randn(32, 2048, 3, device="cuda:0") * 0.5
t = torch.randint(0, 1000, (32,), device="cuda:0")
flat, prep = net.flat_data(), net._prepared()
def folded(): return B.planenet_fwd(flat, x, t, *net.cfg, prepared=prep)[0]
def plain(): return B.planenet_fwd(flat, x, t, *net.cfg, want_stash=True, prepared=prep)[0]
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
with torch.no_grad():
    for _ in range(3): folded(); plain()
    for r in range(4):
        print(f"round {r}: folded {timed(folded):.3f} ms   plain (with stash) {timed(plain):.3f} ms")
