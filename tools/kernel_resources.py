#!/usr/bin/env python3
"""VGPRs / scratch / occupancy of every kernel in the given .hip files (cross-compiles for gfx950, no GPU needed).
usage: tools/kernel_resources.py so3x_resnet.hip so3x_mlp_bwd.hip ...        (no arguments: every kernel source)
tests/test_kernel_resources.py runs scan() over the sources and fails when a hot kernel reports scratch (register spills)."""
import os, re, subprocess, sys, tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-extensions_amd", "csrc")
SOURCES = ("so3x_rotation.hip", "so3x_igso3.hip", "so3x_mlp.hip", "so3x_mlp_bwd.hip", "so3x_train_fused.hip", "so3x_diffusion.hip", "so3x_resnet.hip",
           "so3x_se3.hip", "so3x_stats.hip", "so3x_rotgrad.hip", "so3x_optim.hip", "so3x_protnet.hip", "so3x_protnet_bf16.hip",
           "so3x_planenet.hip", "so3x_planenet_bf16.hip", "so3x_planenet_bf16_bwd.hip")
KEYS = {"VGPRs": r"VGPRs: (\d+)", "AGPRs": r"AGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)",
        "occ": r"Occupancy \[waves/SIMD\]: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)"}
# the build's flags (csrc/Makefile)
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-fast-math", "-fno-slp-vectorize"]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [o if o else n for o, n in zip(out, names)]
    except OSError:
        return list(names)


def scan(files=SOURCES):
    """{demangled kernel name: {"VGPRs", "AGPRs", "scratch", "occ", "lds", "file"}}"""
    res = {}
    for f in files:
        with tempfile.NamedTemporaryFile(suffix=".o") as tmp:
            r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c", f, "-o", tmp.name, "-Rpass-analysis=kernel-resource-usage"],
                               cwd=CSRC, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"hipcc failed on {f}:\n{r.stderr[-2000:]}")
        name, row, found = None, {}, []
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name, row = m.group(1), {}
            for k, pat in KEYS.items():
                m = re.search(pat, line)
                if m:
                    row[k] = int(m.group(1))
            if name and "lds" in row:
                found.append((name, dict(row, file=f)))
                name = None
        for (mangled, row), nice in zip(found, demangle([n for n, _ in found])):
            res[re.sub(r"\(anonymous namespace\)::", "", nice)] = row
    return res


if __name__ == "__main__":
    for name, row in scan(sys.argv[1:] or SOURCES).items():
        short = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"{short[:60]:60s} vgpr {row.get('VGPRs'):3d} agpr {row.get('AGPRs'):3d} scratch {row.get('scratch'):4d} occ {row.get('occ')}")
