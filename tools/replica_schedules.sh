#!/bin/bash
# Recall of N emulated replicas against the one-GPU run as a function of the launches per replica and epoch (Gowalla shape, cap 64):
#   bash tools/replica_schedules.sh  > gpurun_out/replica_schedules.log
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for cfg in "1 12500" "8 1563" "8 3125" "8 6250" "4 3125" "4 6250"; do
  set -- $cfg
  echo "== world $1, $2 users per launch"
  python tools/quality.py --shape gowalla --world $1 --batch $2 --cap 64 --epochs 120 --eval-every 40 2>&1 | grep recall@20 | cut -c1-200
done
