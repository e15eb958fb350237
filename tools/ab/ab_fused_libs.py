#!/usr/bin/env python3
"""A/B of so3x_train_fused ALONE (prep launch + the one kernel; raw C ABI through ctypes) between builds of libso3x.so, interleaved
in one process, 2^19 rotations:   python tools/ab/ab_fused_libs.py build/libso3x_a.so build/libso3x_b.so ... [--json out.json] [--step] [--t-const]
(--step: the whole training step -- + so3x_train_bwd_reduce_adam -- replayed as a captured hipGraph, as bench.py's train_step leg runs it)
(--t-const: every rotation at timestep 500 instead of a drawn one -- the kernel then touches ONE row of the 4 MB inverse-CDF table;
the FETCH_SIZE difference to the default run is what the table's cold fills cost: tools/ab/fused_bounds.sh)
(timing builds made with tools/ab/build_variant.sh <name> "<flags>" so3x_train_fused.hip may compute garbage: only their time counts)"""
import ctypes as C
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
args = sys.argv[1:]
out = args[args.index("--json") + 1] if "--json" in args else None
lg = int(args[args.index("--log2") + 1]) if "--log2" in args else 19
STEP = "--step" in args
TCONST = "--t-const" in args
libs = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] not in ("--json", "--log2"))] or [B.LIB_PATH]
n, T = 1 << lg, 1000
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
proc = SO3Diffusion(net, timesteps=T).to(DEV)
trap_q, _ = proc._tables()
x0 = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
params = net.flat_data().clone()
loss = torch.zeros(1, device=DEV)
t_given = torch.full((n,), 500, dtype=torch.int64, device=DEV) if TCONST else None
ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
calls = {}
for path in libs:
    l = C.CDLL(os.path.abspath(path))
    l.so3x_train_workspace_bytes.restype = C.c_size_t
    ws = torch.empty(int(l.so3x_train_workspace_bytes(C.c_int64(n), C.c_int(T))), dtype=torch.uint8, device=DEV)

    pl, grad = params.clone(), torch.zeros_like(params)
    m, v, stp = torch.zeros_like(params), torch.zeros_like(params), torch.zeros(2, device=DEV)

    def call(l=l, ws=ws, pl=pl, grad=grad, m=m, v=v, stp=stp):
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = l.so3x_train_fused(s, P(pl), P(proc._sched), C.c_int(T), P(trap_q), P(proc._guide_q), P(x0), P(t_given) if TCONST else None, None, C.c_int(1), None, None,
                                C.c_uint64(1), C.c_uint64(0), P(ctr), C.c_int64(0), C.c_int64(n), P(loss), None, None, P(ws), C.c_size_t(ws.numel()))
        assert rc == 0, rc
        if STEP:
            rc = l.so3x_train_bwd_reduce_adam(s, C.c_int64(n), C.c_int(T), None, P(grad), P(ws), C.c_size_t(ws.numel()), P(pl), P(m), P(v), P(stp),
                                              C.c_float(3e-4), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), C.c_float(0.0), C.c_float(1.0))
            assert rc == 0, rc
    if STEP:
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            with torch.cuda.graph(gr, stream=st):
                call()
        calls[path] = gr.replay
    else:
        calls[path] = call
    for _ in range(5):
        call()
torch.cuda.synchronize()
times = {p: [] for p in libs}
for r in range(7):
    for p in libs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            calls[p]()
        a.record()
        for _ in range(50 if STEP else 20):
            calls[p]()
        b.record()
        torch.cuda.synchronize()
        times[p].append(a.elapsed_time(b) / (50 if STEP else 20) * 1e3)
rows = []
for p in libs:
    calls[p]()
    torch.cuda.synchronize()
    rows.append({"build": os.path.basename(p), "us_per_call_median": round(statistics.median(times[p]), 2), "us_min": round(min(times[p]), 2),
                 "loss": float(loss)})
    print(json.dumps(rows[-1]))
if out:
    json.dump({"what": (f"the training step (prep + so3x_train_fused + slab reduction with Adam) as a replayed hipGraph, 2^{lg} rotations, interleaved rounds"
                        if STEP else f"so3x_train_fused alone (prep + kernel), 2^{lg} rotations, interleaved rounds"), "rows": rows}, open(out, "w"), indent=1)
