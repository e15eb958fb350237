"""k_se3_q_sample_target at 2^20 frames: what the per-sample CDF-row gathers cost, and what is left when they cost nothing.
   Same kernel, same inputs, only the timesteps differ: (a) uniform over all 1000 rows (the bench leg: 4 MB of rows + 0.5 MB of
   guides against a 4 MB L2 per XCD), (b) uniform over 100 rows (0.45 MB: L2-resident on every XCD), (c) uniform over 1000 rows
   but SORTED (every wave reads a handful of neighbouring rows), (d) one row for everybody.  python tools/ab/se3_qsample_bound.py [--json out]"""
import sys, os, json, statistics, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.se3 import SE3Diffusion, AffineGrad
out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
dev = "cuda:0"
n = 1 << 20
g = torch.Generator(device=dev).manual_seed(0)
proc = SE3Diffusion(lambda x, t: AffineGrad(x.rot[..., 0], x.shift), timesteps=1000).to(dev)
tq, _ = proc._tables()
xr = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
xs = torch.randn(n, 3, device=dev, generator=g)
t_all = torch.randint(0, 1000, (n,), device=dev, generator=g)
cases = {"(a) t uniform over 1000 rows (the bench leg)": t_all,
         "(b) t uniform over 100 rows (rows + guides 0.45 MB: L2-resident)": torch.randint(450, 550, (n,), device=dev, generator=g),
         "(c) the timesteps of (a), sorted": torch.sort(t_all).values.contiguous(),
         "(d) one row for every frame": torch.full((n,), 500, device=dev, dtype=torch.long)}
o = [torch.empty(n, 3, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)]
P = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
lib = B.lib()


def call(tt):
    rc = lib.so3x_se3_q_sample_target(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(proc._sched), C.c_int(1000), P(tq), P(proc._guide_q),
                                      C.c_float(75.0), P(xr), P(xs), P(tt), C.c_int(1), None, None, None, C.c_uint64(1), C.c_uint64(0), C.c_int64(0),
                                      P(o[0]), P(o[1]), P(o[2]), P(o[3]), C.c_int64(n))
    assert rc == 0, rc


times = {k: [] for k in cases}
for k, tt in cases.items():
    for _ in range(5):
        call(tt)
torch.cuda.synchronize()
for r in range(7):
    for k, tt in cases.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            call(tt)
        b.record(); torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 20 * 1e3)
rows = [{"timesteps": k, "us": round(statistics.median(v), 2), "frac_of_8TBs": round(128 * n / (statistics.median(v) * 1e-6) / 8e12, 4)} for k, v in times.items()]
for r in rows:
    print(json.dumps(r))
if out:
    json.dump({"what": "k_se3_q_sample_target, 2^20 frames, 128 B per frame algorithmic, interleaved rounds (tools/ab/se3_qsample_bound.py): the kernel "
                       "with and without the cost of its per-sample CDF-row gathers", "rows": rows}, open(out, "w"), indent=1)
