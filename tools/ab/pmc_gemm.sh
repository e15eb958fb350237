cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_gemm/$tag -- $R/build/gemm_bf16_test 65536 1536 512 0 > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/pmc_gemm/*/*/*counter_collection.csv')):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gemm256' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in d.items(): print(k, len(v), sum(v)/len(v))
PY
