#!/usr/bin/env python3
"""Condenses gpurun_out/profile_<tag>/ (tools/profile_round.sh) into the tracked profiles/ directory:
   <tag>_bench_line.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_<counters>.csv (per-kernel averages) and
   pmc_traffic.json (what bench.py quotes as roofline.traffic).  No GPU needed."""
import collections
import statistics
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"profile_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, f"{tag}_bench_line.json"))
shutil.copy(glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))

# k_p_sample_chain runs in bench.py with FOUR launch shapes on the same (persistent) grid: 100 steps per launch at 2^20 samples
# (ramp, warm-up, timed region, roofline leg), 1000 steps (full_chain), ONE step at 2^20 (external_loop: 1000 calls -- the majority
# of its dispatches since round 5) and one step at 2^22 (external_loop.at_batch_4M).  The grid is the same for all of them (one
# workgroup per CU), so a dispatch's shape is read off its DURATION, in bins a factor > 2 apart (55 us per step at 2^20 samples):
#   < 0.15 ms: 1 step, 2^20 | 0.15 - 1 ms: 1 step, 2^22 | 2.5 - 15 ms: 100 steps | > 30 ms: 1000 steps | else: not counted.
# Never "the majority", never a plain median over all dispatches (round 5's summary did that and priced one-step counters as a
# 100-step launch).
def chain_steps_class(dur_ms, grid=None):
    if dur_ms is None:
        return None
    if dur_ms < 0.15:
        return 1
    if dur_ms < 1.0:
        return "1@4M"
    if 2.5 <= dur_ms <= 15.0:
        return 100
    if dur_ms > 30.0:
        return 1000
    return None


def chain_is_bf16(kname):
    """k_p_sample_chain<PREC, FAST, PAIR, WIDE, F16>: the headline is PREC = 1 (SO3X_PREC_BF16) with F16 = false"""
    import re
    m = re.search(r"k_p_sample_chainILi(\d)E(?:Lb\dE){3}Lb(\d)E", kname) or re.search(r"k_p_sample_chain<(\d), \w+, \w+, \w+, (\w+)>", kname)
    return bool(m) and m.group(1) == "1" and m.group(2) in ("0", "false")


trace = glob.glob(os.path.join(src, "stats", "*kernel_trace.csv"))
if trace:
    recs = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0) or (1 << 20))
            for r in csv.DictReader(open(trace[0])) if "k_p_sample_chain" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for d, g in recs:
        by[chain_steps_class(d)].append(d)
    hundred = sorted(by.get(100, []))
    json.dump({"what": "k_p_sample_chain dispatch durations (ms) under rocprofv3 --kernel-trace, same command as the bench line; launches classified "
                       "by duration into 1 / 100 / 1000 steps per launch (tools/summarize_profiles.py:chain_steps_class)",
               "dispatches": len(recs), "dispatches_by_steps_per_launch": {str(k): len(v) for k, v in by.items()},
               "ms_100_step_launches_median": statistics.median(hundred) if hundred else None,
               "ms_100_step_launches_mean": sum(hundred) / len(hundred) if hundred else None, "n_100_step_launches": len(hundred),
               "ms_100_step_launches": [round(d, 4) for d in hundred],
               "ms_one_step_launches_median": statistics.median(by[1]) if by.get(1) else None,
               "ms_1000_step_launches": [round(d, 3) for d in by.get(1000, [])]},
              open(os.path.join(dst, f"{tag}_chain_dispatches.json"), "w"), indent=1)

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_kernels import KERNELS
per = {}  # counter -> kernel -> list of per-dispatch values (summed over the agent's instances)
unclassified = collections.Counter()   # counter -> k_p_sample_chain dispatches without a duration (dropped)
legs = {}  # counter -> the k_logprob_score dispatches at 2^20 evaluations in dispatch order, split into bench.py's three legs
for d in sorted(glob.glob(os.path.join(src, "pmc_*/"))):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not f:
        continue
    disp = collections.defaultdict(float)
    meta = {}
    dur = {}   # dispatch id -> ms, from the counter file's own timestamps or the pass's kernel trace (tools/profile_round.sh keeps both)
    for tf in glob.glob(os.path.join(d, "*kernel_trace.csv")):
        for r in csv.DictReader(open(tf)):
            if "Dispatch_Id" in r and "End_Timestamp" in r:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for r in csv.DictReader(open(f[0])):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        disp[key] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0))
        if r.get("End_Timestamp") and r["Dispatch_Id"] not in dur:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for (did, cname), v in disp.items():
        kname, grid = meta[did]
        short = next((k for k in KERNELS if k in kname), None)
        if short == "k_p_sample_chain":
            # the launch shape by duration; a dispatch whose duration is unknown is NOT counted (no guess by majority)
            cls = chain_steps_class(dur.get(did), grid)
            unclassified[cname] += cls is None
            if cls is None:
                continue
            short = {1: "k_p_sample_chain:1step", "1@4M": "k_p_sample_chain:1step_4M", 100: "k_p_sample_chain", 1000: "k_p_sample_chain:1000step"}[cls]
            if not chain_is_bf16(kname):      # the f16 / fp32 operand instantiations: their own rows
                short += ":not_bf16"
        if short:
            per.setdefault(cname, {}).setdefault((short, grid), []).append(v)
    # bench.py times k_logprob_score at 2^20 evaluations on three inputs, in this order: eps from the schedule (config 2b),
    # [the 2^24 run], scalar eps (2a), eps ~ U(0.1, 1); same kernel, same grid -- told apart by dispatch order
    lp = sorted((int(did), grid) for did, (kname, grid) in meta.items() if "k_logprob_score" in kname)
    if lp:
        small = min(g for _, g in lp)
        big_ids = [d for d, g in lp if g != small]
        first_big, last_big = (min(big_ids), max(big_ids)) if big_ids else (1 << 62, 1 << 62)
        after = [d for d, g in lp if g == small and d > last_big]
        groups = {"schedule": [d for d, g in lp if g == small and d < first_big], "scalar": after[:len(after) // 2], "uniform": after[len(after) // 2:]}
        for cname in set(c for _, c in disp):
            for leg, ids in groups.items():
                vals = [disp[(str(d), cname)] for d in ids if (str(d), cname) in disp]
                if vals:
                    legs.setdefault(cname, {})[leg] = statistics.median(vals)
rows = []
for cname, ks in sorted(per.items()):
    for (k, grid), vals in sorted(ks.items()):
        # "mean" = the MEDIAN over the dispatches of ONE launch shape (k_p_sample_chain's shapes are separate rows, see above)
        rows.append({"counter": cname, "kernel": k, "grid_size": grid, "dispatches": len(vals), "mean": statistics.median(vals),
                     "min": min(vals), "max": max(vals)})
if not rows:
    # (a round whose counter passes died: the bench line and the kernel statistics above are copied, the committed counter summaries
    #  stay as they were -- bench.py goes on checking their digests against the sources)
    sys.exit(f"{src}: no counter files -- {tag}_pmc_summary.csv and pmc_traffic.json left untouched")
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)


# PlaneNet's kernels run at two shapes in bench.py (32 x 256 and 32 x 2048 points): the counters quoted are the LARGEST grid's
# (the 2048-point shape; for the persistent GEMM every shape has the same grid and the median is over its four uses)
LARGEST_GRID = ("k_gemm256_bf16", "k_gemm_bf16", "k_gemm_tn", "k_attn_fwd", "k_attn_bwd_dq", "k_attn_bwd_dkv", "k_ln_bf16", "k_ln_bwd_bf16")


def mean(counter, kernel, grid=None):
    c = [r for r in rows if r["counter"] == counter and r["kernel"] == kernel and (grid is None or r["grid_size"] == grid)]
    if not c:
        return None
    if grid is None and kernel in LARGEST_GRID:
        return max(c, key=lambda r: r["grid_size"])["mean"]
    return max(c, key=lambda r: r["dispatches"])["mean"] if grid is None else c[0]["mean"]


line = json.load(open(os.path.join(src, "bench_line.json")))
n = line["config"]["batch_per_gpu"]
spl = int(line["roofline"]["steps_per_launch"])
traffic = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --no-cpu-baseline --steps 300 "
                   f"--warmup 100` ({tag}); counters are KB per dispatch; FETCH_SIZE is doubled (gfx950 reports half of wide "
                   "streaming reads, MI355X_MICROARCH.md HBM section), WRITE_SIZE as is"}
fs, wsz = mean("FETCH_SIZE", "k_p_sample_chain"), mean("WRITE_SIZE", "k_p_sample_chain")
if fs is not None and wsz is not None:
    traffic["k_p_sample_chain"] = {"config": {"batch": n, "steps_per_launch": spl, "precision": line["dtype"]},
                                   "fetch_size_kb": fs, "write_size_kb": wsz, "hbm_bytes_per_launch": int((2 * fs + wsz) * 1024),
                                   "algorithmic_bytes_per_launch": 72 * n,
                                   "dispatches_in_the_median": max(r["dispatches"] for r in rows if r["counter"] == "FETCH_SIZE" and r["kernel"] == "k_p_sample_chain")}
fs1, wsz1 = mean("FETCH_SIZE", "k_p_sample_chain:1step"), mean("WRITE_SIZE", "k_p_sample_chain:1step")
if fs1 is not None and wsz1 is not None:   # the one-call-per-step loop's launches (bench.py external_loop): their own record
    traffic["k_p_sample_chain:1step"] = {"config": {"batch": n, "steps_per_launch": 1, "precision": line["dtype"]}, "fetch_size_kb": fs1,
                                         "write_size_kb": wsz1, "hbm_bytes_per_launch": int((2 * fs1 + wsz1) * 1024), "algorithmic_bytes_per_launch": 72 * n}
lg = [r for r in rows if r["counter"] == "FETCH_SIZE" and r["kernel"] == "k_logprob_score"]
if lg:
    small = min(lg, key=lambda r: r["grid_size"])
    big = max(lg, key=lambda r: r["grid_size"])
    def rec(r):
        w_ = mean("WRITE_SIZE", "k_logprob_score", r["grid_size"])
        return {"fetch_size_kb": r["mean"], "write_size_kb": w_, "hbm_bytes_per_launch": int((2 * r["mean"] + w_) * 1024)}
    traffic["k_logprob_score"] = dict(config={"n": 1 << 20}, algorithmic_bytes_per_launch=56 << 20, **rec(small))
    if big is not small:
        traffic["k_logprob_score"]["at_n_2p24"] = dict(algorithmic_bytes_per_launch=56 << 24, **rec(big))
    for leg, nbytes in (("schedule", 56), ("scalar", 52), ("uniform", 56)):   # per input of bench.py's igso3_eval legs
        f_, w_ = legs.get("FETCH_SIZE", {}).get(leg), legs.get("WRITE_SIZE", {}).get(leg)
        if f_ is not None and w_ is not None:
            traffic["k_logprob_score:" + leg] = {"config": {"n": 1 << 20, "eps_input": leg}, "algorithmic_bytes_per_launch": nbytes << 20,
                                                 "fetch_size_kb": f_, "write_size_kb": w_, "hbm_bytes_per_launch": int((2 * f_ + w_) * 1024)}
for k, cfg, alg in (("k_train_fused", {"n": 1 << 19}, 36 * (1 << 19) + 256 * 17556 * 4), ("k_rigid_move", {"structures": 4096, "residues": 256}, 96 * 4096 * 256), ("k_se3_q_sample_target", {"n": 1 << 20}, 128 << 20),
                    ("k_q_sample_target", {"n": 1 << 19}, 92 << 19), ("k_mlp_fwd_stash", {"n": 1 << 19}, None), ("k_bwd_fused", {"n": 1 << 19}, None),
                    # the wide network's training kernels at 2^19 samples: 512 B per sample and dumped stream (X_l: 7, Y_l: 6, dZ_l: 6 + the head's one tile)
                    ("k_resnet_fwd", {"n": 1 << 19, "what": "training forward: X and Y dumps written"}, 13 * 512 << 19),
                    ("k_resnet_bwd", {"n": 1 << 19, "what": "dX chain: Y read, dZ written"}, (12 * 512 + 64) << 19),
                    ("k_resnet_dw", {"n": 1 << 19, "what": "dW GEMM: X and dZ read"}, (13 * 512 + 64) << 19)):
    f_, w_ = mean("FETCH_SIZE", k), mean("WRITE_SIZE", k)
    if f_ is not None and w_ is not None:
        traffic[k] = {"config": cfg, "algorithmic_bytes_per_launch": alg, "fetch_size_kb": f_, "write_size_kb": w_,
                      "hbm_bytes_per_launch": int((2 * f_ + w_) * 1024)}
# PlaneNet at 32 clouds x 2048 points (N = 65536 tokens): the attention forward reads Q, K, V once and writes O (+ the log-sum-exp)
N_TOK = 32 * 2048
for k, cfg, alg in (("k_attn_fwd", {"clouds": 32, "points": 2048}, N_TOK * (3 * 512 * 2 + 512 * 2 + 4 * 4)),
                    ("k_ln_bf16", {"rows": N_TOK}, N_TOK * (512 * 2 * 2 + 8))):
    f_, w_ = mean("FETCH_SIZE", k), mean("WRITE_SIZE", k)
    if f_ is not None and w_ is not None:
        traffic[k] = {"config": cfg, "algorithmic_bytes_per_launch": alg, "fetch_size_kb": f_, "write_size_kb": w_,
                      "hbm_bytes_per_launch": int((2 * f_ + w_) * 1024)}
# matrix-pipe utilisation of the two chain kernels.  SQ_VALU_MFMA_BUSY_CYCLES sums the cycles each SIMD's matrix pipe is
# executing (= MFMAs x 32 for v_mfma_f32_32x32x16_bf16, checked against SQ_INSTS_VALU_MFMA_MOPS_BF16); SQ_BUSY_CYCLES is
# the kernel's duration in shader cycles counted once per shader engine (32 on this chip: 8 XCDs x 4).
util = {}
for k in ("k_p_sample_chain", "k_resnet_chain", "k_train_fused", "k_bwd_fused", "k_mlp_fwd_stash", "k_q_sample_target",
          "k_gemm256_bf16", "k_gemm_bf16", "k_attn_fwd", "k_gemm_tn", "k_attn_bwd_dq", "k_attn_bwd_dkv"):
    busy, sqb = mean("SQ_VALU_MFMA_BUSY_CYCLES", k), mean("SQ_BUSY_CYCLES", k)
    if busy and sqb:
        util[k] = {"SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES": sqb, "simds": 1024, "shader_engines": 32,
                   "mfma_pipe_busy_frac": busy / (1024.0 * sqb / 32.0),
                   # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md cycle-constants table) summed over waves
                   "valu_busy_frac": (4.0 * mean("SQ_ACTIVE_INST_VALU", k) / (1024.0 * sqb / 32.0)) if mean("SQ_ACTIVE_INST_VALU", k) else None,
                   "SQ_ACTIVE_INST_VALU": mean("SQ_ACTIVE_INST_VALU", k), "SQ_INSTS_VALU": mean("SQ_INSTS_VALU", k),
                   "SQ_WAIT_ANY": mean("SQ_WAIT_ANY", k), "SQ_WAVE_CYCLES": mean("SQ_WAVE_CYCLES", k),
                   "SQ_INSTS_VALU_MFMA_MOPS_BF16": mean("SQ_INSTS_VALU_MFMA_MOPS_BF16", k),
                   "SQ_INSTS_LDS": mean("SQ_INSTS_LDS", k), "SQ_LDS_BANK_CONFLICT": mean("SQ_LDS_BANK_CONFLICT", k),
                   "SQ_LDS_IDX_ACTIVE": mean("SQ_LDS_IDX_ACTIVE", k),
                   "note": "fraction of SIMD-cycles the matrix pipe is executing, at the clock the chip actually holds under this load"}
traffic["mfma_utilisation"] = util
# provenance: which sources the counters were taken on (digest recorded on the GPU box by tools/profile_round.sh) and where the
# summary was made; bench.py withholds every PMC-derived field once the kernel sources no longer match
import subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_digest import digest, unit_digests
try:
    raw = open(os.path.join(src, "csrc_sha256.txt")).read().strip()
    rec = json.loads(raw) if raw.startswith("{") else {"all": raw, "units": None}
except OSError:
    rec = {"all": None, "units": None}
prof_digest = rec["all"]
git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
traffic["_meta"] = {"chain_dispatches_without_a_duration_dropped": dict(unclassified), "tag": tag, "csrc_sha256": prof_digest,
                    "csrc_sha256_by_unit": rec["units"], "csrc_sha256_at_summary": digest(ROOT), "git": git,
                    "note": "csrc_sha256 / csrc_sha256_by_unit = tools/csrc_digest.py on the GPU box at profile time; bench.py quotes a kernel's "
                            "PMC fields only while the digest of the translation unit that kernel is compiled from still matches"}
# instruction mix of the chain kernel's step loop, from the same sources (no GPU needed): what bench.py's vector-port accounting
# takes its MFMA / transcendental counts from
try:
    from count_isa import loop_mix
    cnt, groups = loop_mix("so3x_diffusion.hip", "k_p_sample_chainILi1E", depth=1)
    json.dump({"what": "instruction mix of k_p_sample_chain<bf16>'s per-step loop (tools/count_isa.py, LOOP_DEPTH=1) on the sources of "
                       "csrc_sha256_at_summary", "csrc_sha256": digest(ROOT), "per_64_sample_wave_step": dict(groups),
               "mfma_per_wave_step": groups.get("mfma"), "trans_per_wave_step": groups.get("trans"), "top": dict(cnt.most_common(40))},
              open(os.path.join(dst, f"{tag}_chain_isa_mix.json"), "w"), indent=1)
except Exception as e:  # noqa: BLE001
    print("isa mix failed:", e)
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1)[:3000])
