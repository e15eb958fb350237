"""k_q_sample_target at 2^19 samples with per-sample timesteps (the training step's noising launch), in-tree library and
every build/libso3x_qs*.so, interleaved rounds:   python tools/ab/ab_qsample.py [rounds=9]"""
import sys, os, glob, statistics, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dev = "cuda:0"
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
trap_q, _ = proc._tables()
n = 1 << 19
x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev))
libs = {"in_tree": C.CDLL(B.LIB_PATH)}
for path in sorted(glob.glob(os.path.join(ROOT, "build", "libso3x_qs*.so"))):
    libs[os.path.basename(path)[len("libso3x_"):-3]] = C.CDLL(path)
P = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
x_t, tg, td = torch.empty(n, 3, 3, device=dev), torch.empty(n, 3, device=dev), torch.randint(0, 1000, (n,), device=dev)
outs = {}


def call(lib):
    rc = lib.so3x_q_sample_target(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(proc._sched), C.c_int(1000), P(trap_q), P(proc._guide_q),
                                  P(x0), P(td), C.c_int(1), None, None, None, C.c_uint64(3), C.c_uint64(0), None, C.c_int64(0), P(x_t), P(tg),
                                  None, C.c_int64(n))
    assert rc == 0, rc


for name, lib in libs.items():
    for _ in range(5):
        call(lib)
    torch.cuda.synchronize()
    outs[name] = (x_t.clone(), tg.clone(), td.clone())
times = {k: [] for k in libs}
for r in range(rounds):
    for name, lib in libs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call(lib)
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for name in libs:
    same = all(torch.equal(a, b) for a, b in zip(outs[name], outs["in_tree"]))
    print(f"{name:16s} {statistics.median(times[name]):7.2f} us median {min(times[name]):7.2f} min   same bits {same}")
