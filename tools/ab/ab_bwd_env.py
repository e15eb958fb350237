"""A/B of the stashed fused backward between the launcher's variants (SO3X_AB_BWD unset = shipped, or a value such as `sym`),
interleaved in ONE process:   python tools/ab/ab_bwd_env.py variant [log2_batch=19] [rounds=9] [out.json]
For each: us per mlp_bwd call (backward kernel + k_bwd_reduce), and the gradient's distance from the fp32 backward relative to
its norm at full and ragged sizes."""
import sys, os, json, statistics
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict

var = sys.argv[1]
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 19
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 9
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
params = net.flat_data()
VARIANTS = {"shipped": None, var: var}


def setenv(v):
    os.environ.pop("SO3X_AB_BWD", None)
    if v:
        os.environ["SO3X_AB_BWD"] = v


def case(n):
    g = torch.Generator(device=dev).manual_seed(n)
    x = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
    t = torch.randint(0, 1000, (n,), device=dev, generator=g)
    dout = torch.randn(n, 3, device=dev, generator=g) / n
    out, stash = B.mlp_fwd_stash(params, x, t, 1000)
    return x, t, dout, stash


rows = []
for n in (1, 31, 33, 255, 257, 1000, 4097, 1 << 16, (1 << lg) - 5, 1 << lg):
    x, t, dout, stash = case(n)
    ref = B.mlp_bwd(params, x, t, dout, B.PREC_F32, 1000)
    rec = {"n": n}
    for name, v in VARIANTS.items():
        setenv(v)
        g = B.mlp_bwd(params, x, t, dout, B.PREC_BF16, 1000, zstash=stash)
        rec[name + "_rel_err_vs_fp32"] = float((g - ref).norm() / ref.norm())
        rec[name + "_finite"] = bool(torch.isfinite(g).all())
    setenv(None)
    rows.append(rec)
    print(rec)
n = 1 << lg
x, t, dout, stash = case(n)
times = {k: [] for k in VARIANTS}
for v in VARIANTS.values():
    setenv(v)
    for _ in range(5):
        B.mlp_bwd(params, x, t, dout, B.PREC_BF16, 1000, zstash=stash)
torch.cuda.synchronize()
for r in range(rounds):
    for name, v in VARIANTS.items():
        setenv(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            B.mlp_bwd(params, x, t, dout, B.PREC_BF16, 1000, zstash=stash)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) * 100.0)
setenv(None)
timing = {k: {"us_median": round(statistics.median(v), 2), "us_min": round(min(v), 2)} for k, v in times.items()}
print(timing)
if len(sys.argv) > 4:
    json.dump({"what": f"mlp_bwd with the stash at n = 2^{lg} (prep + backward kernel + slab reduction), 10 calls per timing, interleaved rounds in one process",
               "timing": timing, "parity": rows}, open(sys.argv[4], "w"), indent=1)
