"""A/B of the headline kernel's variants, interleaved in ONE process (cdna_hip_programming.md 5.4 rule 24):
   python tools/ab/ab_chain.py [rounds=7] [out.json]
Variants are selected per launch through the environment switches of the A/B build (libso3x_ab.so = csrc/so3x_diffusion.hip
compiled with -DSO3X_AB_BUILD; the product library has none) or by swapping in an older build of the library (build/libso3x_r02a.so, if present):
   base      bf16 chain kernel as shipped: 2-instruction SiLU from the lane-replicated LDS table, hardware sine / cosine, the
             wave's two tiles as one software-pipelined stream, 8-wave workgroups
   narrow_tab SO3X_AB_TAB=narrow the 2 KB table with its shift-add addressing (3 instructions; bit-identical results)
   cdf_global SO3X_AB_CDF=global the inverse-CDF search of the reverse step on global memory instead of the LDS-staged record
   cw        SO3X_AB_TRIG=cw     Cody-Waite sincos_cw in the reverse step (round 1's trigonometry)
   unpaired  SO3X_AB_PAIR=0      one tile after the other (forward_tile twice)
   blockNNN  SO3X_AB_BLOCK=NNN   other workgroup sizes (256: two per CU; 512: one per CU)
   r02a_*    the build before the table became the product path: exp2 + rcp SiLU, and its 4-instruction table variant
For each: ms per 100-step launch at B = 2^20 (median and min over the rounds), sample-steps/s, and max |x - x_fp32| after
ONE reverse step from the same state with the same Philox noise (the fp32 kernel is the parity path pinned to the oracle)."""
import sys, os, json, statistics
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
_, trap_p = proc._tables()
params = net.flat_data()
n = 1 << 20
x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
VARIANTS = {"base": {}, "narrow_tab": {"SO3X_AB_TAB": "narrow"}, "cw": {"SO3X_AB_TRIG": "cw"}, "unpaired": {"SO3X_AB_PAIR": "0"}, "block256": {"SO3X_AB_BLOCK": "256"},
            "cdf_global": {"SO3X_AB_CDF": "global"}}
# builds of earlier states of the kernel (same ABI), if present: "LIB" = path of the alternative libso3x.so
import glob
for path in sorted(glob.glob(os.path.join(ROOT, "build", "libso3x_*.so"))):
    name = os.path.basename(path)[len("libso3x_"):-3]
    if name == "r02a":
        VARIANTS["r02a_exp2_rcp_silu"] = {"LIB": path}
        VARIANTS["r02a_table_4instr"] = {"LIB": path, "SO3X_AB_SILU": "table"}
    elif not name.startswith("bwd"):
        VARIANTS["lib_" + name] = {"LIB": path}
import ctypes as C
_libs = {None: C.CDLL(os.path.join(os.path.dirname(B.LIB_PATH), "libso3x_ab.so"))}  # the A/B build: the only one with the switches
_cur = [None]


def setenv(env):
    for k in ("SO3X_AB_TRIG", "SO3X_AB_SILU", "SO3X_AB_BLOCK", "SO3X_AB_PAIR", "SO3X_AB_TAB", "SO3X_AB_CDF"):
        os.environ.pop(k, None)
    os.environ.update({k: v for k, v in env.items() if k != "LIB"})
    path = env.get("LIB")
    if path not in _libs:
        _libs[path] = C.CDLL(path)
    _cur[0] = path


def run(steps=100, t0=600, prec=1, xin=None):
    """the chain launch straight through the C ABI of the selected build (the torch operator library is linked to the in-tree one)"""
    lib = _libs[_cur[0]]
    lib.so3x_p_sample_workspace_bytes.restype = C.c_size_t
    xi = x if xin is None else xin
    out = torch.empty_like(xi)
    nb = lib.so3x_p_sample_workspace_bytes(C.c_int(1000), C.c_int(prec))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.so3x_p_sample_chain(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(params), P(proc._sched), C.c_int(1000), P(trap_p),
                                 P(proc._guide_p), P(xi), P(out), C.c_int(t0), C.c_int(steps), None, None, C.c_uint64(1), C.c_uint64(0),
                                 C.c_int64(0), C.c_int64(xi.numel() // 9), C.c_int(prec), P(ws), C.c_size_t(nb))
    assert rc == 0, rc
    return out


for env in VARIANTS.values():  # warm every variant (attribute queries, code load) and ramp the clock
    setenv(env)
    for _ in range(3):
        run()
torch.cuda.synchronize()
times = {k: [] for k in VARIANTS}
for r in range(rounds):
    for name, env in VARIANTS.items():
        setenv(env)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1))
setenv({})
same = bool(torch.equal(run(20, 600), (setenv(VARIANTS["narrow_tab"]), run(20, 600))[1]))
print("wide table vs narrow table after 20 steps at B = 2^20: bit-identical =", same)
assert same
setenv({})
same = bool(torch.equal(run(20, 600), (setenv(VARIANTS["cdf_global"]), run(20, 600))[1]))
print("CDF row staged in LDS vs read from global memory, 20 steps: bit-identical =", same)
assert same
setenv({})
xs = x[:65536].contiguous()
ref_s = run(1, 600, prec=0, xin=xs)
rows = []
for name, env in VARIANTS.items():
    setenv(env)
    d = (run(1, 600, prec=1, xin=xs) - ref_s).abs().reshape(-1, 9).max(1).values
    med, mn = statistics.median(times[name]), min(times[name])
    rows.append({"variant": name, "env": env, "ms_per_100_step_launch_median": round(med, 4), "ms_min": round(mn, 4),
                 "sample_steps_per_s_median": n * 100 / med * 1e3,
                 "one_step_abs_dx_vs_fp32_kernel": {"median": float(d.median()), "p99": float(d.quantile(0.99)), "max": float(d.max())}})
    print(f"{name:20s} median {med:8.4f} ms  min {mn:8.4f} ms  {n * 100 / med * 1e3:.4g} sample-steps/s   |dx| vs fp32: median "
          f"{float(d.median()):.2e} p99 {float(d.quantile(0.99)):.2e} max {float(d.max()):.2e}")
setenv({})
if len(sys.argv) > 2:
    json.dump({"what": "k_p_sample_chain<bf16> variants, B = 2^20, 100 steps per launch, t = 600..501, interleaved rounds in one process",
               "rounds": rounds, "rows": rows}, open(sys.argv[2], "w"), indent=1)
