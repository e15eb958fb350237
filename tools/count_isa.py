#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop from its gfx950 assembly (no GPU needed):
     tools/count_isa.py so3x_diffusion.hip k_p_sample_chainILi1E
compiles the file with the build's flags, finds the kernel whose mangled name contains the pattern, takes the LARGEST
backward-branch region (the per-step loop of the chain kernels) and prints instruction counts by class
(LOOP_DEPTH=1 in the environment: the largest loop nested inside it)."""
import collections, os, re, subprocess, sys, tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-extensions_amd", "csrc")
TRANS = {"v_exp_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_exp_f16", "v_rcp_f16"}


def loop_mix(src, pattern, extra=(), depth=0):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-fast-math",
                        "-fno-slp-vectorize", "-S", "--cuda-device-only", src, "-o", out, *extra], cwd=CSRC, check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pattern) + r"\w*:", l)][0]
    end = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end") and i > start][0]
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    # loops by their head label, outermost first; each with its farthest back-branch.  depth 0 = the largest loop,
    # depth 1 = the largest loop nested inside it (the per-step loop of the chain kernels inside their chunk loop), ...
    heads = {}
    for n, a, b in loops:
        heads[a] = max(heads.get(a, 0), b)
    order = sorted(heads.items(), key=lambda ab: ab[0] - ab[1])
    a, b = order[0]
    for _ in range(depth):
        inner = [(x, y) for x, y in order if x > a and y <= b]
        a, b = inner[0]
    cnt = collections.Counter()
    for l in body[a:b]:
        l = l.strip()
        if l and not l.startswith((".", ";", "/")) and not l.endswith(":"):
            cnt[l.split()[0]] += 1
    groups = collections.Counter()
    for op, c in cnt.items():
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        g = ("mfma" if op.startswith("v_mfma") else "trans" if base in TRANS else "valu" if op.startswith("v_") else
             "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
             "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        groups[g] += c
    return cnt, groups


if __name__ == "__main__":
    depth = int(os.environ.get("LOOP_DEPTH", "0"))
    cnt, groups = loop_mix(sys.argv[1], sys.argv[2], sys.argv[3:], depth)
    print("total", sum(cnt.values()), dict(groups))
    for op, c in cnt.most_common(50):
        print(f"  {op:28s} {c}")
