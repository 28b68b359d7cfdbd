#!/bin/bash
# Per-kernel rocprofv3 times of one launch size (quick A/B of a kernel change; run through gpurun from the repo root):
#   tools/kprof.sh <tag> [launch sizes ...]      -> gpurun_out/kprof_<tag>.txt (kernel, calls, avg us), env (POI_TE_DBG ...) is passed through
set -u
TAG=${1:-x}; shift
SIZES=${*:-12500}
OUT=gpurun_out/kprof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o k -- python tools/launch_sizes.py $SIZES > $OUT/run.log 2>&1
python - "$OUT" <<'PY' > gpurun_out/kprof_$TAG.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:45]:
    print("%-90s %6s calls  avg %8.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
grep launch_users $OUT/run.log >> gpurun_out/kprof_$TAG.txt
rm -rf $OUT
