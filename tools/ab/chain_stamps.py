"""Where a wave's time goes in the bf16 chain kernel: runs a TIMING build (hipcc -DSO3X_STAMPS=1 of so3x_diffusion.hip linked with
the other objects into build/stamps_libso3x.so) whose wave 0 of workgroup 0 accumulates s_memtime intervals per phase of a
step:   python tools/ab/chain_stamps.py [steps=100]"""
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
proc = SO3Diffusion(net, timesteps=1000).to(dev)
_, trap_p = proc._tables()
params = net.flat_data()
n = 1 << 20
x = B.quat_to_rmat(torch.randn(n, 4, device=dev))
lib = C.CDLL(os.path.join(ROOT, "build", "stamps_libso3x.so"))
lib.so3x_p_sample_workspace_bytes.restype = C.c_size_t
nb = lib.so3x_p_sample_workspace_bytes(C.c_int(1000), C.c_int(1))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
names = ["step top (rmat, scalar loads, DMA issue, prefetch)", "the two head stages", "output exchange (bpermute)", "reverse step (incl. vmcnt wait)",
         "layer 0 of both tiles + activation A0", "the six 15-MFMA stages"]
for rep in range(3):
    out = torch.empty_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.so3x_p_sample_chain(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(params), P(proc._sched), C.c_int(1000), P(trap_p),
                                 P(proc._guide_p), P(x), P(out), C.c_int(600), C.c_int(steps), None, None, C.c_uint64(1), C.c_uint64(0),
                                 C.c_int64(0), C.c_int64(n), C.c_int(1), P(ws), C.c_size_t(nb))
    e1.record(); torch.cuda.synchronize()
    assert rc == 0
    acc = out.flatten()[:12].view(torch.int64).tolist()
    tot = sum(acc)
    print(f"launch {rep}: {e0.elapsed_time(e1):.3f} ms; wave 0 of workgroup 0, its first chunk: {tot / steps:.0f} s_memtime ticks per step")
    for k in (0, 4, 5, 1, 2, 3):
        print(f"   {names[k]:55s} {acc[k] / steps:9.0f} ticks per step  {100 * acc[k] / tot:5.1f} %")
