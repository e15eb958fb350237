"""which piece of the prevstep training step fails under hipGraph capture (debug helper)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diffusion-extensions_amd"))
from so3x import backend as B
from so3x.diffusion import SO3Diffusion
from so3x.so3_train import RotPredict
DEV = "cuda:0"
which = sys.argv[1]
torch.manual_seed(0)
n = 4096
x = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
net = RotPredict(out_type="rotmat", precision="bf16").to(DEV)
proc = SO3Diffusion(net, timesteps=100, loss_type="prevstep").to(DEV)
proc.rng_counter = torch.zeros(1, dtype=torch.int64, device=DEV)
t = torch.randint(0, 100, (n,), device=DEV)
x6 = torch.randn(n, 6, device=DEV, requires_grad=True)


def piece():
    if which == "loss":
        xr = B.quat_to_rmat(torch.randn(n, 4, device=DEV)).requires_grad_(True)
        l = B.prevstep_loss(proc._sched, xr, x, x, t)
        l.backward()
        return l
    if which == "six":
        r = B.six2rmat(x6)
        r.sum().backward()
        return r
    if which == "qsample":
        trap_q, _ = proc._tables()
        return B.q_sample_target(proc._sched, trap_q, x, t, want_target=False, rng_offset_dev=proc.rng_counter, guide_q=proc._guide_q)[0]
    if which == "fwd":
        with torch.no_grad():
            return net(x, t, t_table=100)
    if which == "fwdbwd":
        o = net(x, t, t_table=100)
        o.sum().backward()
        return o
    if which == "step":
        l = proc(x)
        l.backward()
        return l


if which.startswith("opt"):
    import faulthandler
    faulthandler.enable()
    from so3x.graphs import TrainStepGraph
    proc.rng_counter = None
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, fused=True, capturable=True)
    if which == "opteager":
        for _ in range(5):
            opt.zero_grad(set_to_none=True)
            loss = proc(x)
            loss.backward()
            opt.step()
        print("eager steps ok", float(loss), flush=True)
    gr = TrainStepGraph(proc, opt, x.shape, warmup=2)
    print("captured", flush=True)
    print([float(gr.step(x)) for _ in range(3)], flush=True)
    sys.exit(0)

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        piece()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print(which, "eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = piece()
print(which, "captured", flush=True)
g.replay()
torch.cuda.synchronize()
print(which, "replayed", float(out.float().sum()), flush=True)
