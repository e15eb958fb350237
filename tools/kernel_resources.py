#!/usr/bin/env python3
"""VGPRs / scratch / occupancy of every kernel in the given .hip files (cross-compiles for gfx950, no GPU needed).
usage: tools/kernel_resources.py so3x_resnet.hip so3x_mlp_bwd.hip ..."""
import os, re, subprocess, sys, tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-extensions_amd", "csrc")
KEYS = {"VGPRs": r"VGPRs: (\d+)", "AGPRs": r"AGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)",
        "occ": r"Occupancy \[waves/SIMD\]: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)"}
for f in sys.argv[1:]:
    with tempfile.NamedTemporaryFile(suffix=".o") as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-fast-math",
                            "-fno-slp-vectorize", "-c", f, "-o", tmp.name, "-Rpass-analysis=kernel-resource-usage"],
                           cwd=CSRC, capture_output=True, text=True)
    name, row = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name, row = m.group(1), {}
        for k, pat in KEYS.items():
            m = re.search(pat, line)
            if m:
                row[k] = int(m.group(1))
        if name and "lds" in row:
            print(f"{name[:100]:100s} vgpr {row.get('VGPRs'):3d} agpr {row.get('AGPRs'):3d} scratch {row.get('scratch'):4d} occ {row.get('occ')}")
            name = None
