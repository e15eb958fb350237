"""k_se3_q_sample_target at 2^20 frames (bench.py's se3 leg) across builds of libso3x.so, interleaved, raw C ABI:
   python tools/ab/ab_se3_qsample.py build/libso3x_a.so ... [--json out.json]"""
import sys, os, json, statistics, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.se3 import SE3Diffusion, AffineGrad
args = sys.argv[1:]
out = args[args.index("--json") + 1] if "--json" in args else None
libs = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] != "--json")] or [B.LIB_PATH]
dev = "cuda:0"
n = 1 << 20
g = torch.Generator(device=dev).manual_seed(0)
proc = SE3Diffusion(lambda x, t: AffineGrad(x.rot[..., 0], x.shift), timesteps=1000).to(dev)
tq, _ = proc._tables()
xr = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
xs = torch.randn(n, 3, device=dev, generator=g)
tt = torch.randint(0, 1000, (n,), device=dev, generator=g)
o = [torch.empty(n, 3, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)]
P = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
handles = {p: C.CDLL(os.path.abspath(p)) for p in libs}


def call(l):
    rc = l.so3x_se3_q_sample_target(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(proc._sched), C.c_int(1000), P(tq), P(proc._guide_q),
                                    C.c_float(75.0), P(xr), P(xs), P(tt), C.c_int(1), None, None, None, C.c_uint64(1), C.c_uint64(0), C.c_int64(0),
                                    P(o[0]), P(o[1]), P(o[2]), P(o[3]), C.c_int64(n))
    assert rc == 0, rc


times = {p: [] for p in libs}
ref = None
same = {}
for p in libs:
    for _ in range(5):
        call(handles[p])
    torch.cuda.synchronize()
    cur = [t.clone() for t in o]
    if ref is None:
        ref = cur
    same[p] = max(float((a - b).abs().max()) for a, b in zip(cur, ref))
for r in range(7):
    for p in libs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            call(handles[p])
        b.record(); torch.cuda.synchronize()
        times[p].append(a.elapsed_time(b) / 20 * 1e3)
rows = [{"build": os.path.basename(p), "us": round(statistics.median(times[p]), 2), "frac_of_8TBs": round(128 * n / (statistics.median(times[p]) * 1e-6) / 8e12, 4),
         "max_abs_diff_vs_first": same[p]} for p in libs]
for r in rows:
    print(json.dumps(r))
if out:
    json.dump({"what": "k_se3_q_sample_target, 2^20 frames, 128 B per frame algorithmic, interleaved rounds (tools/ab/ab_se3_qsample.py)", "rows": rows}, open(out, "w"), indent=1)
