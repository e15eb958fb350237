#!/usr/bin/env python3
"""Phase stamps of the one-kernel training step (timing build: tools/ab/build_variant.sh tf_stamps "-DTF_STAMPS" so3x_train_fused.hip):
where a round's cycles go in workgroup 0's chain wave 0 and dW wave 0.   python tools/ab/fused_stamps.py build/libso3x_tf_stamps.so [out.json]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-extensions_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from so3x import backend as B  # noqa: E402
from so3x.diffusion import SO3Diffusion  # noqa: E402
from so3x.so3_train import RotPredict  # noqa: E402

DEV = "cuda:0"
n, T = 1 << 19, 1000
torch.manual_seed(0)
net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
proc = SO3Diffusion(net, timesteps=T).to(DEV)
trap_q, _ = proc._tables()
x0 = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
params = net.flat_data().clone()
loss = torch.zeros(1, device=DEV)
out = torch.zeros(n, 3, device=DEV)
l = C.CDLL(os.path.abspath(sys.argv[1]))
l.so3x_train_workspace_bytes.restype = C.c_size_t
ws = torch.empty(int(l.so3x_train_workspace_bytes(C.c_int64(n), C.c_int(T))), dtype=torch.uint8, device=DEV)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(4):
    out.zero_()
    ev0.record()
    rc = l.so3x_train_fused(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(params), P(proc._sched), C.c_int(T), P(trap_q), P(proc._guide_q),
                            P(x0), None, None, C.c_int(1), None, None, C.c_uint64(1), C.c_uint64(it), None, C.c_int64(0), C.c_int64(n), P(loss), None,
                            P(out), P(ws), C.c_size_t(ws.numel()))
    assert rc == 0
    ev1.record()
    torch.cuda.synchronize()
    call_us = ev0.elapsed_time(ev1) * 1e3
st = out.reshape(-1)[:2 * 2 * 64 * 32].view(torch.int64).cpu().numpy().reshape(2, 64, 32)
rounds = 16
names_c = ["top"] + ["l0+mfma", "act0", "hidden", "head+loss"] + sum([[f"wait_done{l}", f"store{l}", f"dh{l}" if l else "wait_handed", f"end{l}"] for l in (4, 3, 2, 1, 0)], [])
res = {"call_us_with_stamps": call_us}
c = st[0, :rounds, :25].astype(np.float64)
d = np.diff(c, axis=1)                     # phases 0->1 ... 23->24
nxt = c[1:, 0] - c[:-1, 24]                # end of round -> next top
res["chain_wave0_cycles_per_phase_median"] = {nm: float(np.median(d[1:-1, i])) for i, nm in enumerate(names_c[1:])}
res["chain_round_cycles_median"] = float(np.median(c[2:, 0] - c[1:-1, 0]))
e = st[:, 63, :6].astype(np.float64)
res["dw_wave0_ticks_entry_to_first_round_to_slab_to_slab_done_to_exit"] = [float(st[1, 0, 0] - e[1, 0]), float(e[1, 4] - st[1, 0, 0]), float(e[1, 5] - e[1, 4]), float(e[1, 1] - e[1, 5])]
res["kernel_ticks_entry_to_exit"] = [float(e[0, 1] - e[0, 0]), float(e[1, 1] - e[1, 0])]
res["kernel_us_by_the_100MHz_clock"] = [float(e[0, 3] - e[0, 2]) / 100.0, float(e[1, 3] - e[1, 2]) / 100.0]
res["shader_ticks_per_us"] = float(e[0, 1] - e[0, 0]) / (float(e[0, 3] - e[0, 2]) / 100.0)
res["prologue_ticks_chain"] = float(c[0, 0] - e[0, 0])
res["epilogue_ticks_chain"] = float(e[0, 1] - c[rounds - 1, 24])
res["chain_between_rounds"] = float(np.median(nxt))
res["chain_loop_span_cycles"] = float(c[rounds - 1, 24] - c[0, 0])
res["chain_round_starts"] = [float(x - c[0, 0]) for x in c[:, 0]]
res["chain_forward_cycles_by_round"] = [float(x) for x in (c[:, 4] - c[:, 0])]     # even rounds carry the dW waves' noise pass on the same SIMDs
res["chain_backward_cycles_by_round"] = [float(x) for x in (c[:, 24] - c[:, 4])]
w = st[1, :rounds, :17].astype(np.float64)
order = [0] + sum([[2 + 3 * k, 3 + 3 * k, 4 + 3 * k] for k in range(4)], []) + [14, 15, 1, 16]   # ... B1(0), products(0), noise pass, B2(0)
names_w = sum([[f"hand{l}", f"products{l}", f"end{l}"] for l in (4, 3, 2, 1)], []) + ["hand0", "products0", "noise", "end0"]
seq = w[:, order]
dw = np.diff(seq, axis=1)
res["dw_wave0_even_rounds"] = {nm: float(np.median(dw[2:-1:2, i])) for i, nm in enumerate(names_w)}
res["dw_wave0_odd_rounds"] = {nm: float(np.median(dw[1:-1:2, i])) for i, nm in enumerate(names_w)}
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
