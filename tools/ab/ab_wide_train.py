"""forward-with-stash and backward of the wide network timed separately (C ABI through the binding, best of 5)"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
# (the operators come from libso3x_torch.so, which finds libso3x.so beside itself: to compare builds, copy one over
#  diffusion-extensions_amd/libso3x.so on the GPU box between two runs of this script; the argument is only a label)
dev = "cuda:0"
torch.manual_seed(0)
pw = torch.randn(B.N_PARAMS_RESNET, device=dev) * 0.06
n = 1 << 19
x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
t = torch.randint(0, 1000, (n,), device=dev)
dout = torch.randn(n, 3, device=dev)


def best(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    b = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        b = min(b, e0.elapsed_time(e1))
    return b


out, stash = B.resnet_fwd_stash(pw, x, t, 1000, 1)
print(sys.argv[1:] or "in-tree", "fwd_stash ms %.4f" % best(lambda: B.resnet_fwd_stash(pw, x, t, 1000, 1)),
      "bwd(stash) ms %.4f" % best(lambda: B.resnet_bwd(pw, x, t, dout, 1000, 1, stash=stash)),
      "fwd ms %.4f" % best(lambda: B.resnet_fwd(pw, x, t, 1000, 1)))
