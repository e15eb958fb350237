#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (development aid; results go to stdout as JSON lines)."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion

if os.environ.get("SO3X_LIB"):  # A/B: an alternative build of the same ABI
    B.LIB_PATH = os.path.abspath(os.environ["SO3X_LIB"])
dev = "cuda:0"


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    which = set(sys.argv[1:]) or {"logprob", "rot", "chain", "train", "se3"}
    g = torch.Generator(device=dev).manual_seed(0)
    if "logprob" in which:
        for lg in (20, 24):
            n = 1 << lg
            R = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
            eps = torch.rand(n, device=dev, generator=g) * 0.9 + 0.1
            ms = timeit(lambda: B.igso3_logprob_score(R, eps))
            print(json.dumps({"k": "logprob_score", "n": n, "ms": ms, "GBs": 56 * n / ms / 1e6, "frac8T": 56 * n / ms / 1e6 / 8000}))
    if "rot" in which:
        n = 1 << 22
        R = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
        k = torch.rand(n, device=dev, generator=g)
        ms = timeit(lambda: B.so3_scale(R, k))
        print(json.dumps({"k": "so3_scale", "n": n, "ms": ms, "GBs": 76 * n / ms / 1e6}))
        ms = timeit(lambda: B.quat_to_rmat(torch.empty(0, 4, device=dev)) if False else B.log_rmat_vec(R))
        print(json.dumps({"k": "log_rmat_vec", "n": n, "ms": ms, "GBs": 48 * n / ms / 1e6}))
    if "igsample" in which:
        from so3x.distributions import IsotropicGaussianSO3
        for lg in (20, 22):
            n = 1 << lg
            d1 = IsotropicGaussianSO3(torch.tensor(0.5, device=dev))                      # one shared CDF row (p_sample's case)
            ms = timeit(lambda: d1.sample((n,)))
            print(json.dumps({"k": "igso3_sample_scalar_eps", "n": n, "ms": ms, "samples_per_s": n / ms * 1e3, "GBs": 36 * n / ms / 1e6}))
        n = 1 << 20
        sched = torch.from_numpy(B.schedule_from_betas(B.cosine_beta_schedule(1000))).to(dev)
        trap = B.igso3_build_tables(sched[4])
        guide = B.igso3_build_guide(trap)
        t = torch.randint(0, 1000, (n,), device=dev)
        for gname, gd in (("noguide", None), ("guide", guide)):
            ms = timeit(lambda: B.igso3_sample(trap, n, row_idx=t, seed=1, guide=gd) if gd is not None else B.igso3_sample(trap, n, row_idx=t, seed=1))
            print(json.dumps({"k": "igso3_sample_per_sample_eps_" + gname, "n": n, "ms": ms, "samples_per_s": n / ms * 1e3, "GBs": 44 * n / ms / 1e6}))
    if "rotgrad" in which:
        n = 1 << 22
        x6 = torch.randn(n, 6, device=dev, generator=g)
        G = torch.randn(n, 3, 3, device=dev, generator=g)
        R = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
        R2 = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
        R3 = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
        t = torch.randint(0, 1000, (n,), device=dev)
        sched = torch.from_numpy(B.schedule_from_betas(B.cosine_beta_schedule(1000))).to(dev)
        gd = torch.randn(n, device=dev, generator=g)
        for name, fn, nbytes in (("six2rmat", lambda: B.six2rmat(x6), 24 + 36),
                                 ("six2rmat_bwd", lambda: B._Six2Rmat.backward(type("c", (), {"saved_tensors": (x6,)})(), G), 24 + 36 + 24),
                                 ("log_rmat_bwd", lambda: B.log_rmat_bwd(R, G), 36 * 3),
                                 ("rmat_dist_bwd", lambda: B.rmat_dist_bwd(R, R2, gd), 36 * 4 + 4),
                                 ("prevstep_loss+grad", lambda: B.prevstep_loss(sched, R, R2, R3, t), 36 * 4 + 8)):
            ms = timeit(fn)
            print(json.dumps({"k": name, "n": n, "ms": ms, "GBs": nbytes * n / ms / 1e6, "frac8T": nbytes * n / ms / 1e6 / 8000}))
    if "se3" in which:
        from so3x.se3 import SE3Diffusion, AffineGrad
        S, L = 4096, 256
        rot = B.quat_to_rmat(torch.randn(S, 4, device=dev, generator=g))
        shift = torch.randn(S, 3, device=dev, generator=g)
        pos = torch.randn(S, L, 3, device=dev, generator=g)
        frames = B.quat_to_rmat(torch.randn(S, L, 4, device=dev, generator=g))
        ms = timeit(lambda: B.rigid_move(rot, shift, pos, frames))
        print(json.dumps({"k": "rigid_move", "residues": S * L, "ms": ms, "GBs": 96 * S * L / ms / 1e6, "frac8T": 96 * S * L / ms / 1e6 / 8000}))
        n = 1 << 20
        proc3 = SE3Diffusion(lambda x, t: AffineGrad(x.rot[..., 0], x.shift), timesteps=1000).to(dev)
        tq, _ = proc3._tables()
        xr = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
        xs = torch.randn(n, 3, device=dev, generator=g)
        tt = torch.randint(0, 1000, (n,), device=dev, generator=g)
        ms = timeit(lambda: B.se3_q_sample_target(proc3._sched, tq, 75.0, xr, xs, tt, seed=1))
        print(json.dumps({"k": "se3_q_sample_target", "n": n, "ms": ms, "GBs": (84 + 48 + 8) * n / ms / 1e6}))
    if "qsample" in which:
        from so3x.so3_train import RotPredict as _RP
        from so3x.diffusion import SO3Diffusion as _SD
        pr = _SD(_RP(out_type="skewvec"), timesteps=1000).to(dev)
        tq, _ = pr._tables()
        for lg in (19, 22):
            n = 1 << lg
            x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
            tt = torch.randint(0, 1000, (n,), device=dev, generator=g)
            ms = timeit(lambda: B.q_sample_target(pr._sched, tq, x0, tt, seed=1))
            print(json.dumps({"k": "q_sample_target", "n": n, "ms": ms, "GBs": 92 * n / ms / 1e6, "frac8T": 92 * n / ms / 1e6 / 8000}))
            ms = timeit(lambda: B.q_sample_target(pr._sched, tq, x0, tt, seed=1, guide_q=pr._guide_q))
            print(json.dumps({"k": "q_sample_target_guided", "n": n, "ms": ms, "GBs": 92 * n / ms / 1e6, "frac8T": 92 * n / ms / 1e6 / 8000}))
    if "chain" in which or "train" in which:
        torch.manual_seed(0)
        net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
        proc = SO3Diffusion(net, timesteps=1000).to(dev)
        _, trap_p = proc._tables()
        params = net.flat_params_nograd()
    if "chain" in which:
        n = 1 << 20
        x = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
        for prec, name in ((1, "bf16"), (0, "fp32")):
            steps = 50 if prec else 10
            ms = timeit(lambda: B.p_sample_chain(params, proc._sched, trap_p, x, 600, steps, seed=1, precision=prec), reps=3, warm=1)
            print(json.dumps({"k": f"chain_{name}", "n": n, "steps": steps, "ms": ms, "sample_steps_per_s": n * steps / ms * 1e3,
                              "mfma_algo_TFLOPs": 34190 * n * steps / ms / 1e9}))
        ms = timeit(lambda: B.p_sample_chain(params, proc._sched, trap_p, x, 600, 1, seed=1, precision=1), reps=10)
        print(json.dumps({"k": "step_bf16", "n": n, "ms": ms, "sample_steps_per_s": n / ms * 1e3}))
    if "resnet" in which or "resnetchain" in which:
        FLOP = 781830  # 6 x 2 x 255 x 255 + 6 x 255 (bias adds folded: counted as mul-add) ... algorithmic, per sample
        pw = torch.randn(B.N_PARAMS_RESNET, device=dev, generator=g) * 0.06
        proc_t = SO3Diffusion(RotPredict(out_type="skewvec"), timesteps=1000).to(dev)
        _, trap_p = proc_t._tables()
        for lg in (16, 18, 20) if "resnet" in which else ():
            n = 1 << lg
            x = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
            t = torch.randint(0, 1000, (n,), device=dev)
            ms = timeit(lambda: B.resnet_fwd(pw, x, t, 1000, precision=1), reps=5, warm=2)
            print(json.dumps({"k": "resnet_fwd_bf16", "n": n, "ms": ms, "algo_TFLOPs": FLOP * n / ms / 1e9}))
        if "resnet" in which:
            n = 1 << 14
            x = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
            t = torch.randint(0, 1000, (n,), device=dev)
            ms = timeit(lambda: B.resnet_fwd(pw, x, t, 1000, precision=0), reps=3, warm=1)
            print(json.dumps({"k": "resnet_fwd_fp32", "n": n, "ms": ms, "algo_TFLOPs": FLOP * n / ms / 1e9}))
        for lg, steps in ((16, 100), (18, 50)) if "resnet" in which else ((18, 50),):
            n = 1 << lg
            x = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
            ms = timeit(lambda: B.resnet_p_sample_chain(pw, proc_t._sched, trap_p, x, 600, steps, seed=1, precision=1), reps=3, warm=1)
            print(json.dumps({"k": "resnet_chain_bf16", "n": n, "steps": steps, "ms": ms, "sample_steps_per_s": n * steps / ms * 1e3,
                              "algo_TFLOPs": FLOP * n * steps / ms / 1e9}))
    if "resnettrain" in which:
        from so3x.so3_lock_train import RotPredict as WideNet
        torch.manual_seed(0)
        wnet = WideNet(out_type="skewvec", precision="bf16").to(dev)
        wproc = SO3Diffusion(wnet, timesteps=1000).to(dev)
        opt = torch.optim.Adam(wnet.parameters(), lr=3e-4, fused=True)
        for lg in (15, 19):
            n = 1 << lg
            x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))

            def wstep():
                loss = wproc(x0)
                opt.zero_grad()
                loss.backward()
                opt.step()
            ms = timeit(wstep, reps=5, warm=2)
            print(json.dumps({"k": "resnet_train_step_bf16", "n": n, "ms": ms, "samples_per_s": n / ms * 1e3,
                              "algo_TFLOPs": 3 * 781830 * n / ms / 1e9}))
    if "traingraph" in which:
        from so3x.graphs import TrainStepGraph
        for wide in (False, True):
            torch.manual_seed(0)
            if wide:
                from so3x.so3_lock_train import RotPredict as Net
            else:
                Net = RotPredict
            gnet = Net(out_type="skewvec", precision="bf16").to(dev)
            gproc = SO3Diffusion(gnet, timesteps=1000).to(dev)
            gopt = torch.optim.Adam(gnet.parameters(), lr=3e-4, fused=True, capturable=True)
            n = 1 << 19
            x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
            tg = TrainStepGraph(gproc, gopt, x0.shape)
            ms = timeit(lambda: tg.graph.replay(), reps=10, warm=3)
            print(json.dumps({"k": "train_step_graph_" + ("wide" if wide else "mlp65"), "n": n, "ms": ms, "samples_per_s": n / ms * 1e3,
                              "loss": float(tg.loss)}))
    if "prevstep" in which:
        from so3x.graphs import TrainStepGraph
        torch.manual_seed(0)
        pnet = RotPredict(out_type="rotmat", precision="bf16").to(dev)
        pproc = SO3Diffusion(pnet, timesteps=1000, loss_type="prevstep").to(dev)
        popt = torch.optim.Adam(pnet.parameters(), lr=3e-4, fused=True, capturable=True)
        n = 1 << 19
        x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
        tg = TrainStepGraph(pproc, popt, x0.shape)
        ms = timeit(lambda: tg.graph.replay(), reps=10, warm=3)
        print(json.dumps({"k": "train_step_graph_prevstep_rotmat", "n": n, "ms": ms, "samples_per_s": n / ms * 1e3, "loss": float(tg.loss)}))
    if "train" in which:
        n = 1 << 19
        x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
        for prec in ("bf16", "fp32"):
            net.precision = prec
            opt = torch.optim.Adam(net.parameters(), lr=3e-4, fused=True)

            def step():
                loss = proc(x0)
                opt.zero_grad()
                loss.backward()
                opt.step()
            ms = timeit(step, reps=5, warm=2)
            print(json.dumps({"k": f"train_step_{prec}", "n": n, "ms": ms, "samples_per_s": n / ms * 1e3}))
            t = torch.randint(0, 1000, (n,), device=dev)
            ms = timeit(lambda: B.mlp_fwd(params, x0, t, B.PREC_BF16 if prec == "bf16" else B.PREC_F32), reps=5)
            print(json.dumps({"k": f"mlp_fwd_{prec}", "n": n, "ms": ms, "TFLOPs": 34190 * n / ms / 1e9}))


if __name__ == "__main__":
    main()
