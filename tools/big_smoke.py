"""Large-batch smoke (development aid): 2^28 rotations through the streaming kernels and two chain steps; the tail of every
result must equal the same call on the tail alone (64-bit indexing, grid caps, index_base)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
dev = "cuda:0"
n = (1 << 28) + 12345   # element indices beyond 2^31
q = torch.randn(n, 4, device=dev)
R = B.quat_to_rmat(q); del q
tail = slice(n - 1000, n)
ref = B.quat_to_rmat
k = torch.rand(n, device=dev)
S = B.so3_scale(R, k)
small = B.so3_scale(R[tail].contiguous(), k[tail].contiguous())
assert torch.equal(S[tail], small), "so3_scale tail mismatch"
del S
lp, sc, _ = B.igso3_logprob_score(R, k * 0.9 + 0.1)
lp2, sc2, _ = B.igso3_logprob_score(R[tail].contiguous(), (k * 0.9 + 0.1)[tail].contiguous())
assert torch.equal(lp[tail], lp2) and torch.equal(sc[tail], sc2), "logprob tail mismatch"
del lp, sc
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
tq, tp = proc._tables()
t = torch.randint(0, 1000, (n,), device=dev)
xt, tg, _ = B.q_sample_target(proc._sched, tq, R, t, seed=1, rng_offset=3, guide_q=proc._guide_q)
xt2, tg2, _ = B.q_sample_target(proc._sched, tq, R[tail].contiguous(), t[tail].contiguous(), seed=1, rng_offset=3, index_base=n - 1000,
                                guide_q=proc._guide_q, quirk_col0=False)
xt3, tg3, _ = B.q_sample_target(proc._sched, tq, R, t, seed=1, rng_offset=3, guide_q=proc._guide_q, quirk_col0=False)
assert torch.equal(xt3[tail], xt2) and torch.equal(tg3[tail], tg2), "q_sample_target tail mismatch"
del xt, tg, xt3, tg3
out = B.p_sample_chain(net.flat_params_nograd(), proc._sched, tp, R, 500, 2, seed=2, precision=1, guide_p=proc._guide_p)
out2 = B.p_sample_chain(net.flat_params_nograd(), proc._sched, tp, R[tail].contiguous(), 500, 2, seed=2, precision=1, index_base=n - 1000, guide_p=proc._guide_p)
assert torch.equal(out[tail], out2), "chain tail mismatch"
assert torch.isfinite(out).all()
print("big smoke ok, n =", n, "max mem GB", torch.cuda.max_memory_allocated() / 2**30)
