#!/usr/bin/env python3
"""Digests of the kernel sources: what ties a committed PMC profile (profiles/pmc_traffic.json, `_meta`) to the code it was taken on.

digest():   sha256 over ALL kernel sources (diffusion-extensions_amd/csrc/*.hip, *.hpp, *.inc, *.cpp, Makefile + include/so3x.h), file
            names included.
unit_digests(): one sha256 per translation unit (csrc/*.hip): the unit, every csrc-local header it includes (transitively) and the
            Makefile (compiler flags).  include/so3x.h is left out on purpose: it changes with every new entry point of another unit;
            the constants a kernel takes from it (precision codes, error codes) have been fixed since round 1.
bench.py quotes a kernel's PMC-derived fields only while the digest of THE UNIT THAT KERNEL IS COMPILED FROM still matches the
profile's (KERNEL_UNIT below); tools/profile_round.sh records both forms on the GPU box at profile time (JSON on stdout)."""
import glob
import hashlib
import json
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")

# kernel (as named in profiles/pmc_traffic.json) -> the translation unit it is compiled from
KERNEL_UNIT = {
    "k_p_sample_chain": "so3x_diffusion.hip", "k_logprob_score": "so3x_igso3.hip", "k_train_fused": "so3x_train_fused.hip",
    "k_resnet_chain": "so3x_resnet.hip", "k_resnet_fwd": "so3x_resnet.hip", "k_resnet_bwd": "so3x_resnet.hip", "k_resnet_dw": "so3x_resnet.hip",
    "k_rigid_move": "so3x_se3.hip", "k_se3_q_sample_target": "so3x_se3.hip",
    "k_q_sample_target": "so3x_mlp_bwd.hip", "k_mlp_fwd_stash": "so3x_mlp_bwd.hip", "k_bwd_fused": "so3x_mlp_bwd.hip",
    "k_gemm256_bf16": "so3x_planenet_bf16.hip", "k_gemm_bf16": "so3x_planenet_bf16.hip", "k_attn_fwd": "so3x_planenet_bf16.hip",
    "k_ln_bf16": "so3x_planenet_bf16.hip", "k_gemm_tn": "so3x_planenet_bf16_bwd.hip", "k_attn_bwd_dq": "so3x_planenet_bf16_bwd.hip",
    "k_attn_bwd_dkv": "so3x_planenet_bf16_bwd.hip", "k_ln_bwd_bf16": "so3x_planenet_bf16_bwd.hip",
    "k_prot_encoder": "so3x_protnet.hip", "k_prot_encoder_bwd": "so3x_protnet_bwd.hip",
}


def _csrc(root):
    return os.path.join(root, "diffusion-extensions_amd", "csrc")


def digest(root=ROOT):
    h = hashlib.sha256()
    csrc = _csrc(root)
    files = sorted(f for pat in ("*.hip", "*.hpp", "*.inc", "*.cpp", "Makefile") for f in glob.glob(os.path.join(csrc, pat)))
    files.append(os.path.join(root, "include", "so3x.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _closure(csrc, name, seen):
    if name in seen or not os.path.exists(os.path.join(csrc, name)):
        return
    seen.add(name)
    with open(os.path.join(csrc, name), encoding="utf-8", errors="replace") as fh:
        for inc in re.findall(r'^\s*#\s*include\s+"([^"/]+)"', fh.read(), flags=re.M):
            _closure(csrc, inc, seen)


def unit_digests(root=ROOT):
    csrc = _csrc(root)
    out = {}
    for unit in sorted(os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip"))):
        seen = set()
        _closure(csrc, unit, seen)
        h = hashlib.sha256()
        for name in sorted(seen) + ["Makefile"]:
            h.update(name.encode())
            with open(os.path.join(csrc, name), "rb") as fh:
                h.update(fh.read())
        out[unit] = h.hexdigest()
    return out


def kernel_unit(kernel):
    """the unit of a pmc_traffic.json key ("k_logprob_score:schedule", "k_p_sample_chain:1step" -> their kernel's)"""
    return KERNEL_UNIT.get(kernel.split(":")[0])


if __name__ == "__main__":
    print(json.dumps({"all": digest(), "units": unit_digests()}))
