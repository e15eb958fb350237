#!/usr/bin/env python3
"""What HBM gives a plain streaming kernel on this box: fill (write only), reduction (read only), copy (1 read : 1 write) over 4 GiB with
torch's own kernels -- the practical ceilings the store-bound (wide-net forward dumps, slab writes) and copy-like (q_sample) kernels are
priced against, beside the 8 TB/s of the data sheet.   python tools/ab/hbm_rates.py [out.json]"""
import json
import sys

import torch

DEV = "cuda:0"
n = 1 << 30  # fp32 elements: 4 GiB
a = torch.empty(n, device=DEV)
b = torch.empty(n, device=DEV)
a.normal_()


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); f(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e-3


res = {"bytes": 4 * n}
res["write_only_TBps"] = 4 * n / timed(lambda: b.fill_(1.0)) / 1e12
res["read_only_TBps"] = 4 * n / timed(lambda: a.sum()) / 1e12
res["copy_TBps_read_plus_write"] = 8 * n / timed(lambda: b.copy_(a)) / 1e12
res["add_2reads_1write_TBps"] = 12 * n / timed(lambda: torch.add(a, b, out=b)) / 1e12
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
