#!/usr/bin/env python3
"""The chain kernel's f16-operand leg beside the bf16 headline: B = 2^20, 100 steps per launch, interleaved
rounds in one process, HIP events; and how far each is from the fp32 chain after one step.  usage: ab_chain_f16.py [out.json]"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-extensions_amd")]
import torch  # noqa: E402
from so3x import backend as B  # noqa: E402
from so3x.diffusion import SO3Diffusion  # noqa: E402
from so3x.so3_train import RotPredict  # noqa: E402

DEV = "cuda:0"
n, T = 1 << 20, 1000
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
proc = SO3Diffusion(net, timesteps=T).to(DEV)
_, trap_p = proc._tables()
x = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
params = net.flat_params_nograd()


def launch(prec, xx, t0=T - 1, steps=100, off=0):
    return B.p_sample_chain(params, proc._sched, trap_p, xx, t0, steps, seed=1, rng_offset=off, precision=prec, out=xx, guide_p=proc._guide_p)


for _ in range(8):   # clock ramp
    launch(B.PREC_BF16, x.clone())
times = {1: [], 2: []}
for r in range(9):
    for prec in (1, 2):
        xx = x.clone()
        launch(prec, xx)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(3):
            launch(prec, xx, T - 1 - 100 * i, 100, 100 * i)
        b.record()
        torch.cuda.synchronize()
        times[prec].append(a.elapsed_time(b) / 3)
ref = B.p_sample_chain(params, proc._sched, trap_p, x.clone(), 500, 1, seed=1, precision=B.PREC_F32)
err = {}
for prec in (1, 2):
    out = B.p_sample_chain(params, proc._sched, trap_p, x.clone(), 500, 1, seed=1, precision=prec)
    e = (out - ref).abs().amax(dim=(1, 2))
    err[prec] = {"median": float(e.median()), "p99": float(e.quantile(0.99)), "max": float(e.max())}
rows = []
for prec, name in ((1, "bf16 operands (the headline kernel)"), (2, "f16 operands (SO3X_PREC_F16: extra leg)")):
    ms = statistics.median(times[prec])
    rows.append({"variant": name, "ms_per_100_step_launch": round(ms, 4), "sample_steps_per_s": n * 100 / (ms * 1e-3),
                 "frac_of_bf16_mfma_peak_algorithmic": 34190 * n * 100 / (ms * 1e-3) / 2.5e15,
                 "one_step_at_t500_vs_fp32_chain_max_abs_entry": err[prec]})
res = {"what": "k_p_sample_chain, B = 2^20, 100 steps per launch, interleaved rounds, HIP events (tools/ab/ab_chain_f16.py)", "rows": rows}
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
