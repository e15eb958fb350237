#!/bin/bash
# quick PMC passes over one command: tools/pmc_quick.sh <out-name> <script and args...>   (run through gpurun from the repo root)
name=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$name
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$i -o p -- python3 $R/"$@" > $out/pmc_$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, os, sys, collections, statistics
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "*counter_collection.csv")):
    per = collections.defaultdict(float); meta = {}
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); meta[r["Dispatch_Id"]] = r["Kernel_Name"][:60]
    for (d, c), v in per.items():
        agg[meta[d]][c].append(v)
    os.remove(f)
with open(os.path.join(out, "summary.csv"), "w") as g:
    for k in sorted(agg):
        for c in sorted(agg[k]):
            g.write(f"{k!r},{c},{len(agg[k][c])},{statistics.median(agg[k][c])}\n")
for f in glob.glob(os.path.join(out, "pmc_*", "*kernel_trace.csv")):
    os.remove(f)
PY
grep -i "prot\|k_train_fused\|k_gemm\|k_attn" $out/summary.csv | head -80
