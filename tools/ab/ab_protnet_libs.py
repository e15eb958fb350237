#!/usr/bin/env python3
"""A/B of so3x_protnet_fwd (bf16 form, 4096 complexes x (198 + 58) residues) between builds of libso3x.so, raw C ABI through ctypes,
interleaved in one process:   python tools/ab/ab_protnet_libs.py build/libso3x_a.so build/libso3x_b.so ...
(timing builds made with tools/ab/build_variant.sh <name> "-DPROT_AB_..." so3x_protnet_bf16.hip may compute garbage: only their time counts)"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-extensions_amd")]
import torch  # noqa: E402
from so3x import backend as B  # noqa: E402
from so3x.models import ProtNet  # noqa: E402

DEV = "cuda:0"
libs = [a for a in sys.argv[1:] if not a.startswith("--")] or [B.LIB_PATH]
Bn, lr, ll = 4096, 198, 58
torch.manual_seed(0)
net = ProtNet(precision="bf16").to(DEV).eval()
params = net.flat_params_nograd()
g = torch.Generator(device=DEV).manual_seed(1)


def chains(n, L):
    res = torch.zeros(n * L, 21, device=DEV)
    res[torch.arange(n * L, device=DEV), torch.randint(0, 21, (n * L,), device=DEV, generator=g)] = 1.0
    return res, torch.randn(n * L, 3, device=DEV, generator=g) * 8.0, B.quat_to_rmat(torch.randn(n * L, 4, device=DEV, generator=g)).reshape(n * L, 9), \
        torch.arange(0, n * L + 1, L, device=DEV, dtype=torch.int64)


rec, lig = chains(Bn, lr), chains(Bn, ll)
t = torch.randint(0, 1000, (Bn,), device=DEV, generator=g)
out = torch.empty(Bn, 6, device=DEV)
P = lambda a: C.c_void_p(a.data_ptr())  # noqa: E731
calls = {}
for path in libs:
    l = C.CDLL(os.path.abspath(path))
    l.so3x_protnet_workspace_bytes.restype = C.c_size_t
    nb = int(l.so3x_protnet_workspace_bytes(C.c_int64(Bn), C.c_int64(max(lr, ll)), C.c_int64(Bn * lr), C.c_int64(Bn * ll), C.c_int(64), C.c_int(4), C.c_int(4), C.c_int(3), C.c_int(1)))
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)

    def call(l=l, ws=ws, nb=nb):
        rc = l.so3x_protnet_fwd(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(params), P(rec[0]), P(rec[1]), P(rec[2]), P(rec[3]), C.c_int64(Bn * lr),
                                P(lig[0]), P(lig[1]), P(lig[2]), P(lig[3]), C.c_int64(Bn * ll), P(t), P(out), None, None, C.c_int64(Bn), C.c_int64(max(lr, ll)),
                                C.c_int(64), C.c_int(4), C.c_int(4), C.c_int(3), C.c_int(1), None, P(ws), C.c_size_t(nb), C.c_float(0.0), C.c_uint64(0), C.c_uint64(0))
        assert rc == 0, rc
    calls[path] = call
for c in calls.values():
    for _ in range(3):
        c()
torch.cuda.synchronize()
times = {p: [] for p in calls}
for rnd in range(7):
    for p, c in calls.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            c()
        e1.record()
        torch.cuda.synchronize()
        times[p].append(e0.elapsed_time(e1) / 5)
for p, v in times.items():
    print(f"{os.path.basename(p):40s} median {statistics.median(v):.3f} ms  min {min(v):.3f}")
