#!/usr/bin/env python3
"""sha256 over the kernel sources (diffusion-extensions_amd/csrc/*.hip, *.hpp, *.inc, *.cpp, Makefile + include/so3x.h), file names
included: what ties a committed PMC profile (profiles/pmc_traffic.json, `_meta.csrc_sha256`) to the code it was taken on.  bench.py
quotes PMC-derived fields only while this digest still matches; tools/profile_round.sh records it on the GPU box at profile time."""
import glob
import hashlib
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def digest(root=ROOT):
    h = hashlib.sha256()
    csrc = os.path.join(root, "diffusion-extensions_amd", "csrc")
    files = sorted(f for pat in ("*.hip", "*.hpp", "*.inc", "*.cpp", "Makefile") for f in glob.glob(os.path.join(csrc, pat)))
    files.append(os.path.join(root, "include", "so3x.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(digest())
