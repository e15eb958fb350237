import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-extensions_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from so3x import backend as B
from so3x.diffusion import SO3Diffusion
from so3x.so3_train import RotPredict
from test_train_fused import _fused, _staged
DEV = "cuda:0"
torch.manual_seed(1)
net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
proc = SO3Diffusion(net, timesteps=1000).to(DEV)
for n in (32, 64, 128, 256, 1000):
    x0 = B.quat_to_rmat(torch.randn(n, 4, device=DEV, generator=torch.Generator(device=DEV).manual_seed(n)))
    params = net.flat_data()
    loss, grad, t_used, x_t, out = _fused(B, proc, params, x0, seed=7, rng_offset=3)
    ls, gs, ts, xs, os_ = _staged(B, proc, params, x0, seed=7, rng_offset=3)
    print(n, "loss", float(loss), float(ls), "t eq", bool(torch.equal(ts, t_used)), "x eq", bool(torch.equal(xs, x_t)),
          "out maxdiff", float((out - os_).abs().max()), "grad rel", float((grad - gs).abs().max() / gs.abs().max()))
    d = (out - os_).abs().max(1).values
    bad = (d > 1e-6).nonzero().flatten()[:16].tolist()
    print("   bad rows", bad, out[:2].tolist(), os_[:2].tolist())
    _, tgt, _ = B.q_sample_target(proc._sched, proc._trap_q, x0, t_used, quirk_col0=True, seed=7, rng_offset=3, guide_q=proc._guide_q)
    print("   tg maxdiff", float((out - tgt).abs().max()), out[:3].tolist(), tgt[:3].tolist(), "xdiff", float((x_t - xs).abs().max()))
