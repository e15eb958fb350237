"""Loss and parameter gradient of one training step (so3x_train_fwd + so3x_train_bwd through the C ABI) for the in-tree library
and every build/libso3x_*.so given, on the same inputs and Philox draws:  python tools/ab/cmp_train_grad.py n T lib.so [lib.so ...]"""
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "diffusion-extensions_amd")):
    sys.path.insert(0, os.path.abspath(p))
import torch
from so3x import backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion

n, T = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda:0"
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(dev)
with torch.no_grad():
    for p_ in net.parameters():
        p_.mul_(float(sys.argv[sys.argv.index("--wscale") + 1]) if "--wscale" in sys.argv else 1.0)
proc = SO3Diffusion(net, timesteps=T).to(dev)
trap_q, _ = proc._tables()
params = net.flat_data()
x0 = B.quat_to_rmat(torch.randn(n, 4, device=dev))
P = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
paths = [B.LIB_PATH] + [a for a in sys.argv[3:] if a.endswith(".so")]
res = {}
for path in paths:
    lib = C.CDLL(path)
    lib.so3x_train_workspace_bytes.restype = C.c_size_t
    lib.so3x_mlp_stash_bytes.restype = C.c_size_t
    nb = lib.so3x_train_workspace_bytes(C.c_int64(n), C.c_int(T))
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    st = torch.zeros(lib.so3x_mlp_stash_bytes(C.c_int64(n)), dtype=torch.uint8, device=dev)
    x_t, dout, out = torch.empty(n, 3, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
    t_draw = torch.empty(n, dtype=torch.int64, device=dev)
    loss, grad, one = torch.zeros(1, device=dev), torch.zeros(params.numel(), device=dev), torch.ones(1, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.so3x_train_fwd(s, P(params), P(proc._sched), C.c_int(T), P(trap_q), P(proc._guide_q), P(x0), None, P(t_draw), C.c_int(1), None, None,
                            C.c_uint64(7), C.c_uint64(0), None, C.c_int64(0), C.c_int64(n), P(x_t), P(dout), P(st), P(loss), P(out), P(ws), C.c_size_t(nb))
    assert rc == 0, rc
    rc = lib.so3x_train_bwd(s, P(x_t), P(t_draw), P(dout), P(st), C.c_int64(n), C.c_int(T), P(one), P(grad), P(ws), C.c_size_t(nb))
    assert rc == 0, rc
    torch.cuda.synchronize()
    res[path] = (float(loss), grad.clone(), out.clone(), st.view(torch.float16).float().clone(), dout.clone(), x_t.clone(), t_draw.clone())
ref = res[paths[-1]]
for path in paths:
    l, g, o = res[path][:3]
    print(f"{os.path.basename(path):28s} loss {l:.6f}  |grad| {float(g.norm()):.5e}  rel grad diff vs last {float((g - ref[1]).norm() / ref[1].norm()):.3e}  "
          f"max |out diff| {float((o - ref[2]).abs().max()):.3e}  finite {bool(torch.isfinite(g).all())}")
    zs, zr = res[path][3], ref[3]
    d = (zs - zr).abs()
    print("   stash: max |dz|", float(d.max()), "mean", float(d.mean()), " dout diff", float((res[path][4] - ref[4]).abs().max()), " x_t diff",
          float((res[path][5] - ref[5]).abs().max()), " t same", bool((res[path][6] == ref[6]).all()))
    if float(d.max()) > 0.05:
        idx = int(d.argmax())
        per = 17408 // 2
        print("   worst stash element: tile", idx // per, "offset in tile (halfs)", idx % per, "layer", (idx % per) // 2176, "in-layer", (idx % per) % 2176, zs.flatten()[idx].item(), zr.flatten()[idx].item())
        big = (d > 0.05).nonzero().flatten()
        print("   count > 0.05:", big.numel(), "of", d.numel(), "; in-layer offsets (first 20):", sorted(set(((big % per) % 2176).tolist()))[:20])
