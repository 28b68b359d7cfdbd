#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats, and separate PMC passes for HBM traffic and the SQ counters.
# usage: tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/{stats,fetch,write,sq}
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --full-out /tmp/bench_prof_full.json --steps 2 --warmup 1 --eval-steps 2 --no-cpu-baseline --no-quality --no-secondary --no-exact --no-x1"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o k -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  -f csv -d $OUT/sq -o s -- $CMD > $OUT/sq.log 2>&1
# calibration: pure-MFMA loops under the same SQ counter set (what MFMA_BUSY/BUSY reads at 100 % matrix-pipe issue)
[ -x tools/micro/mfma_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_calib.hip -o tools/micro/mfma_calib > $OUT/calib_build.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  -f csv -d $OUT/calib -o c -- tools/micro/mfma_calib > $OUT/calib.log 2>&1
python bench.py --full-out $OUT/bench_full.json > $OUT/bench.json 2> $OUT/bench.err
ls -R $OUT | head -30
