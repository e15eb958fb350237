#!/bin/bash
# Produces the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01      ->  gpurun_out/profile_r01/{bench_line.json, stats/, pmc_*/}
# then `python tools/summarize_profiles.py r01` (no GPU needed) copies the summaries into profiles/.
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profile_$tag
mkdir -p $out
cd $R
# what the counters below belong to: the digest of the kernel sources AS THEY ARE ON THIS BOX (bench.py quotes PMC fields only
# while profiles/pmc_traffic.json's digest matches the sources it runs with)
python3 tools/csrc_digest.py > $out/csrc_sha256.txt
python3 bench.py > $out/bench_stdout.txt 2> $out/bench_stderr.txt
tail -1 $out/bench_stdout.txt > $out/bench_line.json
cd /tmp; export TMPDIR=/tmp
# (every profiler pass under its own `timeout`: a pass that dies with a malformed-packet error leaves rocprofv3 waiting on its queue
#  for ever -- 44 minutes of a round's GPU budget, once)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o b -- python3 $R/bench.py --no-cpu-baseline > $out/stats.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$i -o p -- python3 $R/bench.py --no-cpu-baseline --steps 300 --warmup 100 > $out/pmc_$i.log 2>&1
done
# keep what tools/summarize_profiles.py reads, in a size gpurun copies back (<= 64 MiB for all of gpurun_out/): the counter files
# with seven columns and kernel names cut to 100 characters, the PMC passes' own kernel traces cut to three columns
python3 - "$out" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
# the PMC passes' kernel traces shrink to (dispatch id, start, end): tools/summarize_profiles.py joins them to the counters to tell
# a kernel's launch shapes apart by DURATION (k_p_sample_chain runs with 1, 100 and 1000 steps per launch on the same grid)
for f in glob.glob(os.path.join(out, "pmc_*", "*kernel_trace.csv")):
    rows = list(csv.DictReader(open(f)))
    cols = [c for c in ("Dispatch_Id", "Start_Timestamp", "End_Timestamp") if rows and c in rows[0]]
    with open(f, "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=cols)
        w.writeheader()
        w.writerows({c: r[c] for c in cols} for r in rows)
# the counter files: ONE row per (dispatch, counter) -- the values of a counter's instances (XCDs, shader engines) added up, which is
# the first thing the summariser does with them -- and only the dispatches of the kernels it reports (tools/profile_kernels.py); the
# transformer legs of bench.py are thousands of small launches that nothing reads
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools"))
from profile_kernels import KERNELS
keep = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
for f in glob.glob(os.path.join(out, "pmc_*", "*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    cols = [k for k in keep if rows and k in rows[0]]
    agg = {}
    for r in rows:
        if not any(k in r["Kernel_Name"] for k in KERNELS):
            continue
        key = (r["Dispatch_Id"], r["Counter_Name"])
        if key in agg:
            agg[key]["Counter_Value"] = repr(float(agg[key]["Counter_Value"]) + float(r["Counter_Value"]))
        else:
            agg[key] = {k: (r[k][:100] if k == "Kernel_Name" else r[k]) for k in cols}
    with open(f, "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=cols)
        w.writeheader()
        w.writerows(agg.values())
for f in glob.glob(os.path.join(out, "stats", "*kernel_trace.csv")):
    rows = list(csv.DictReader(open(f)))
    cols = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size", "Grid_Size_X", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size"]
    cols = [c for c in cols if rows and c in rows[0]]
    with open(f, "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=cols)
        w.writeheader()
        w.writerows({c: (r[c][:100] if c == "Kernel_Name" else r[c]) for c in cols} for r in rows)
PY
du -sh $out
ls $out $out/stats | head -30
cat $out/bench_line.json | cut -c1-600
