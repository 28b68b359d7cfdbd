#!/bin/bash
# usage: tools/ab_eval.sh <POI_SCORE_DBG values...>   - one bench run per value, prints the evaluation timings
for d in "$@"; do
  POI_SCORE_DBG=$d python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | D=$d python -c "
import sys,json,os
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j['eval']
print('dbg', os.environ['D'], 'eval ms', round(e['ms_per_eval'],2), 'score', round(e['ms_score_topk_per_eval'],2), 'predict', round(e['ms_predict_per_eval'],2), 'frac', round(e['score_topk_frac_of_f32_mfma_peak'],3))"
done
