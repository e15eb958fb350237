"""A/B of builds / launcher variants of the headline kernel, interleaved in ONE process (cdna_hip_programming.md 5.4 rule 24):

   python tools/ab/ab_chain_libs.py [--rounds 9] [--json out.json] name=lib.so[:ENV=VAL[,ENV=VAL...]] ...

Every variant is one library (same C ABI; `ab` = diffusion-extensions_amd/libso3x_ab.so, `product` = libso3x.so) plus the
environment switches to set while it launches (the A/B build's SO3X_AB_*).  The FIRST variant is the reference of the bit
comparison.  Per variant: ms per 100-step launch at B = 2^20 (t = 600..501; median and min over the rounds), sample-steps/s,
whether 20 steps reproduce the first variant's bits, max |x - x_fp32| after ONE reverse step from the same state with the same
Philox noise (the fp32 kernel is the parity path pinned to the oracle), and -- for builds that leave the clock words
(so3x_p_sample_clock_offset) -- the in-kernel clock of the timed launches."""
import argparse, ctypes as C, json, os, statistics, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, p)
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--json")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--batch-log2", type=int, default=20)
ap.add_argument("--prepared", action="store_true", help="time so3x_p_sample_prepared (the one-call-per-step loop's launch) instead of the chain entry")
ap.add_argument("--reps", type=int, default=1, help="launches per timed sample")
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

ALIAS = {"ab": os.path.join(os.path.dirname(B.LIB_PATH), "libso3x_ab.so"), "product": B.LIB_PATH}
variants = {}
for spec in args.variants:
    name, rest = spec.split("=", 1)
    path, _, envs = rest.partition(":")
    env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    variants[name] = (os.path.abspath(ALIAS.get(path, path)), env)
AB_KEYS = ("SO3X_AB_TRIG", "SO3X_AB_SILU", "SO3X_AB_BLOCK", "SO3X_AB_PAIR", "SO3X_AB_TAB", "SO3X_AB_CDF")

dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
_, trap_p = proc._tables()
params = net.flat_data()
n = 1 << args.batch_log2
x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
libs, ws_cache = {}, {}


def lib_of(path):
    if path not in libs:
        l = C.CDLL(path)
        l.so3x_p_sample_workspace_bytes.restype = C.c_size_t
        if hasattr(l, "so3x_p_sample_clock_offset"):
            l.so3x_p_sample_clock_offset.restype = C.c_size_t
        libs[path] = l
    return libs[path]


def run(name, steps=None, t0=600, prec=1, xin=None):
    path, env = variants[name]
    for k in AB_KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    lib = lib_of(path)
    xi = x if xin is None else xin
    out = torch.empty_like(xi)
    nb = lib.so3x_p_sample_workspace_bytes(C.c_int(1000), C.c_int(prec))
    ws = ws_cache.setdefault((path, prec), torch.zeros(nb, dtype=torch.uint8, device=dev))
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    if args.prepared:
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if (path, prec, "prepared") not in ws_cache:
            rc = lib.so3x_p_sample_prepare(st, P(params), C.c_int(1000), P(trap_p), P(proc._guide_p), C.c_int(prec), P(ws), C.c_size_t(nb))
            assert rc == 0, (name, rc)
            ws_cache[(path, prec, "prepared")] = True
        rc = lib.so3x_p_sample_prepared(st, P(proc._sched), C.c_int(1000), P(trap_p), P(proc._guide_p), P(xi), P(out), C.c_int(t0), None,
                                        C.c_int(args.steps if steps is None else steps), None, None, C.c_uint64(1), C.c_uint64(0), C.c_int64(0),
                                        C.c_int64(xi.numel() // 9), C.c_int(prec), P(ws), C.c_size_t(nb))
        assert rc == 0, (name, rc)
        return out
    rc = lib.so3x_p_sample_chain(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(params), P(proc._sched), C.c_int(1000), P(trap_p),
                                 P(proc._guide_p), P(xi), P(out), C.c_int(t0), C.c_int(args.steps if steps is None else steps), None, None,
                                 C.c_uint64(1), C.c_uint64(0), C.c_int64(0), C.c_int64(xi.numel() // 9), C.c_int(prec), P(ws), C.c_size_t(nb))
    assert rc == 0, (name, rc)
    return out


def clock_ghz(name, prec=1):
    path, _ = variants[name]
    lib = lib_of(path)
    if not hasattr(lib, "so3x_p_sample_clock_offset"):
        return None
    off = lib.so3x_p_sample_clock_offset(C.c_int(1000), C.c_int(prec))
    w = ws_cache[(path, prec)][off:off + 16].view(torch.int64).tolist()
    return w[0] / w[1] * 0.1 if w[1] > 0 else None


for name in variants:  # warm every variant (attribute queries, code load) and ramp the clock
    for _ in range(4):
        run(name)
torch.cuda.synchronize()
times, clocks = {k: [] for k in variants}, {k: [] for k in variants}
for r in range(args.rounds):
    for name in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run(name)
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / args.reps)
        clocks[name].append(clock_ghz(name))
first = next(iter(variants))
base20 = run(first, 20)
xs = x[:65536].contiguous()
ref_s = run(first, 1, prec=0, xin=xs)
rows = []
for name, (path, env) in variants.items():
    same = bool(torch.equal(run(name, 20), base20))
    d = (run(name, 1, prec=1, xin=xs) - ref_s).abs().reshape(-1, 9).max(1).values
    med, mn = statistics.median(times[name]), min(times[name])
    ck = [c for c in clocks[name] if c]
    rows.append({"variant": name, "lib": os.path.relpath(path, ROOT), "env": env, "ms_per_launch_median": round(med, 4), "ms_min": round(mn, 4),
                 "sample_steps_per_s_median": n * args.steps / med * 1e3, "bits_equal_first_variant_after_20_steps": same,
                 "in_kernel_clock_ghz_median": round(statistics.median(ck), 4) if ck else None,
                 "one_step_abs_dx_vs_fp32_kernel": {"median": float(d.median()), "p99": float(d.quantile(0.99)), "max": float(d.max())}})
    print(f"{name:22s} median {med:8.4f} ms  min {mn:8.4f}  {n * args.steps / med * 1e3:.4g} ss/s  bits==first {str(same):5s} clock "
          f"{rows[-1]['in_kernel_clock_ghz_median']} GHz  |dx| vs fp32: median {float(d.median()):.2e} p99 {float(d.quantile(0.99)):.2e}", flush=True)
if args.json:
    json.dump({"what": f"k_p_sample_chain<bf16> builds / variants, B = 2^{args.batch_log2}, {args.steps} steps per launch, t = 600 down, "
                       "interleaved rounds in one process", "rounds": args.rounds, "rows": rows}, open(args.json, "w"), indent=1)
