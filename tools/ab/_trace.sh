cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/ab/planenet_small_train_trace.py 2>&1 | grep wall
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_x -o a -- python3 $R/tools/ab/planenet_small_train_trace.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/tr_x/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    per=float(r["TotalDurationNs"])/60/1e3
    tot+=per
    if per>15: print(f'{r["Name"][:80]:80s} calls {int(r["Calls"])/60:5.1f} avg {float(r["AverageNs"])/1e3:8.1f} us  per step {per:8.1f} us')
print("kernel time per step us", tot)
PY
rm -rf $R/gpurun_out/tr_x
