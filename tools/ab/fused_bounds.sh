#!/bin/bash
# The two bounds DESIGN.md section 7 states for k_train_fused, measured:   (through gpurun, from the repo root)
#   bash tools/ab/fused_bounds.sh <out-name>
# (a) HBM traffic above the algorithmic bytes = cold fills of the 4 MB inverse-CDF table into eight L2s: FETCH_SIZE with drawn
#     timesteps against FETCH_SIZE with every rotation at ONE timestep (one 4 KB row of the table);
# (b) LDS bank-conflict cycles = the SiLU table's data-dependent gathers: SQ_LDS_BANK_CONFLICT of the shipped kernel against a
#     counters-only build whose gathers are forced onto distinct bank groups (-DFUSED_AB_TAB_SPREAD; same instruction stream).
# Needs $R/build/libso3x_tabspread.so:   tools/ab/build_variant.sh tabspread "-DFUSED_AB_TAB_SPREAD" so3x_train_fused.hip
name=${1:-fused_bounds}
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_quick.sh ${name}_drawn tools/ab/ab_fused_libs.py $R/diffusion-extensions_amd/libso3x.so > /dev/null
bash $R/tools/pmc_quick.sh ${name}_tconst tools/ab/ab_fused_libs.py $R/diffusion-extensions_amd/libso3x.so --t-const > /dev/null
bash $R/tools/pmc_quick.sh ${name}_tabspread tools/ab/ab_fused_libs.py $R/build/libso3x_tabspread.so > /dev/null
python3 - $R/gpurun_out $name <<'PY'
import csv, json, os, sys
root, name = sys.argv[1:3]
res = {}
for leg in ("drawn", "tconst", "tabspread"):
    row = {}
    for r in csv.reader(open(os.path.join(root, f"{name}_{leg}", "summary.csv"))):
        if "k_train_fused" in r[0]:
            row[r[1]] = float(r[3])
    # FETCH_SIZE / WRITE_SIZE count KB; FETCH_SIZE doubled for gfx950 (the guide's correction, as tools/summarize_profiles.py applies it)
    if "FETCH_SIZE" in row:
        row["fetch_MB"] = round(2 * row["FETCH_SIZE"] / 1024, 2)
    if "WRITE_SIZE" in row:
        row["write_MB"] = round(row["WRITE_SIZE"] / 1024, 2)
    if "SQ_LDS_BANK_CONFLICT" in row and "SQ_ACTIVE_INST_LDS" in row:
        row["lds_conflict_frac_of_lds_active"] = round(row["SQ_LDS_BANK_CONFLICT"] / row["SQ_ACTIVE_INST_LDS"], 4)
    res[leg] = row
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(root, f"{name}.json"), "w"), indent=1)
PY
