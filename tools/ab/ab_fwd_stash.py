"""Timing of the training forward (so3x_mlp_fwd_stash: k_prep + k_mlp_fwd_stash) through the C ABI, for the in-tree library and
any build/libso3x_*.so beside it, interleaved in one process:   python tools/ab/ab_fwd_stash.py [log2_batch=19] [rounds=9]"""
import sys, os, glob, statistics, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 9
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
params = net.flat_data()
n = 1 << lg
x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
t = torch.randint(0, 1000, (n,), device=dev)
libs = {"in_tree": C.CDLL(B.LIB_PATH)}
for path in sorted(glob.glob(os.path.join(ROOT, "build", "libso3x_fwd*.so"))):
    libs[os.path.basename(path)[len("libso3x_"):-3]] = C.CDLL(path)
P = lambda a: C.c_void_p(a.data_ptr())


def run(lib):
    lib.so3x_mlp_workspace_bytes.restype = C.c_size_t
    lib.so3x_mlp_stash_bytes.restype = C.c_size_t
    nb = lib.so3x_mlp_workspace_bytes(C.c_int64(n), C.c_int(1), C.c_int(1000))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = torch.empty(lib.so3x_mlp_stash_bytes(C.c_int64(n)), dtype=torch.uint8, device=dev)
    out = torch.empty(n, 3, device=dev)
    def call():
        rc = lib.so3x_mlp_fwd_stash(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(params), P(x), P(t), C.c_int64(1), P(out), P(st),
                                    C.c_int64(n), C.c_int(3), C.c_int(1), C.c_int(1000), P(ws), C.c_size_t(nb))
        assert rc == 0, rc
    return call, out


calls = {k: run(v) for k, v in libs.items()}
for c, _ in calls.values():
    for _ in range(5):
        c()
torch.cuda.synchronize()
times = {k: [] for k in calls}
for r in range(rounds):
    for k, (c, _) in calls.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c()
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) * 100.0)
ref = calls["in_tree"][1]
for k, v in times.items():
    d = (calls[k][1] - ref).abs().max().item()
    print(f"{k:24s} {statistics.median(v):8.2f} us median  {min(v):8.2f} us min per call (prep + forward), n = 2^{lg};  max |out - in_tree| = {d:.3e}")
