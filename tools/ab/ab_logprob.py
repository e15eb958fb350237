"""k_logprob_score at 2^20 and 2^24 evaluations (BASELINE config 2) for the in-tree library and every build/libso3x_lps*.so,
graph-replayed launches timed with events, interleaved rounds in one process:   python tools/ab/ab_logprob.py [rounds=7]"""
import sys, os, glob, statistics, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda:0")
libs = {"in_tree": C.CDLL(B.LIB_PATH)}
for path in sorted(glob.glob(os.path.join(ROOT, "build", "libso3x_lps*.so"))):
    libs[os.path.basename(path)[len("libso3x_"):-3]] = C.CDLL(path)
for lg in (20, 24):
    n = 1 << lg
    g = torch.Generator(device=dev).manual_seed(0)
    R = B.quat_to_rmat(torch.randn(n, 4, device=dev, generator=g))
    eps = torch.rand(n, device=dev, generator=g) * 0.9 + 0.1
    logp, score = torch.empty(n, device=dev), torch.empty(n, 3, device=dev)
    reps = 50 if lg == 20 else 10
    graphs, outs = {}, {}
    for name, lib in libs.items():
        def launch(lib=lib):
            rc = lib.so3x_igso3_logprob_score(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(R.data_ptr()), C.c_void_p(eps.data_ptr()),
                                              C.c_int64(1), C.c_void_p(logp.data_ptr()), C.c_void_p(score.data_ptr()), None, C.c_int64(n))
            assert rc == 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                launch()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                launch()
        gr.replay()
        torch.cuda.synchronize()
        graphs[name] = gr
        outs[name] = (logp.clone(), score.clone())
    times = {k: [] for k in libs}
    for r in range(rounds):
        for name, gr in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / reps * 1e3)
    for name in libs:
        us = statistics.median(times[name])
        same = torch.equal(outs[name][0], outs["in_tree"][0]) and torch.equal(outs[name][1], outs["in_tree"][1])
        print(f"n=2^{lg} {name:12s} {us:8.2f} us median {min(times[name]):8.2f} min  {56 * n / us / 1e6:6.2f} TB/s = {56 * n / us / 1e6 / 8 * 100:5.1f} % of 8 TB/s  same bits {same}")
