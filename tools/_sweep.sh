cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_q -o k -- python bench.py --steps 3 --warmup 1 --no-quality --no-secondary --no-cpu-baseline --no-eval > gpurun_out/prof_q.log 2>&1
python - <<EOP
import csv,glob
f=glob.glob("gpurun_out/prof_q/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("te_reduce","te_hot","te_d","te_gather","rs_","te_segment","te_rowmap","te_finalize","dense_apply","te_parts")):
        print("%-60s %4s %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3))
EOP
