"""Determinism soak (development aid): the kernels with intra-workgroup hand-offs (LDS rings, chain/dW wave roles, counted
waits) are run over and over on the same inputs and must reproduce their first result bit for bit -- a synchronisation slip
shows up as a sporadic mismatch.   python tools/soak.py [iterations]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.so3_lock_train import RotPredict as Wide
from so3x.diffusion import SO3Diffusion

iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
wide = Wide(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=200).to(dev)
_, trap_p = proc._tables()
trap_q = proc._trap_q
p, pw = net.flat_params_nograd(), wide.flat_params_nograd()
n = 100003  # ragged on purpose
x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
t = torch.randint(0, 200, (n,), device=dev)
dout = torch.randn(n, 3, device=dev)
cases = {
    "chain bf16": lambda: B.p_sample_chain(p, proc._sched, trap_p, x, 199, 20, seed=1, precision=1, guide_p=proc._guide_p),
    "chain fp32": lambda: B.p_sample_chain(p, proc._sched, trap_p, x[:20000].contiguous(), 199, 5, seed=1, precision=0),
    "wide chain bf16": lambda: B.resnet_p_sample_chain(pw, proc._sched, trap_p, x, 199, 4, seed=1, precision=1, guide_p=proc._guide_p),
    "mlp fwd_stash+bwd": lambda: B.mlp_bwd(p, x, t, dout, 1, 200, zstash=B.mlp_fwd_stash(p, x, t, 200)[1]),
    "mlp bwd recompute": lambda: B.mlp_bwd(p, x, t, dout, 1, 200),
    "mlp bwd fp32": lambda: B.mlp_bwd(p, x[:30000].contiguous(), t[:30000].contiguous(), dout[:30000].contiguous(), 0, 200),
    "wide fwd_stash+bwd": lambda: B.resnet_bwd(pw, x, t, dout, 200, 1, stash=B.resnet_fwd_stash(pw, x, t, 200, 1)[1]),
    "wide bwd fp32": lambda: B.resnet_bwd(pw, x[:20000].contiguous(), t[:20000].contiguous(), dout[:20000].contiguous(), 200, 0),
    "q_sample_target": lambda: torch.cat([v.reshape(-1) for v in B.q_sample_target(proc._sched, trap_q, x, t, seed=3, guide_q=proc._guide_q)[:2]]),
}
contend = "--contend" in sys.argv  # a second stream keeps the memory system and the CUs busy: timing changes, results must not
side = torch.cuda.Stream()
junk = torch.randn(1 << 26, device=dev)
bad = 0
for name, fn in cases.items():
    ref = fn().clone()
    mism = 0
    for i in range(iters):
        if contend:
            with torch.cuda.stream(side):
                for _ in range(3):
                    junk.mul_(1.0001)
                    B.quat_to_rmat(junk[: 1 << 22].view(-1, 4))
        if not torch.equal(fn(), ref):
            mism += 1
    bad += mism
    print(f"{name:22s} {iters} runs, {mism} mismatches", flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
