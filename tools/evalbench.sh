#!/bin/bash
# Short timing of the evaluation pass on the GPU box (through gpurun): users/s, ms per evaluation, scoring kernel share.
timeout 400 python bench.py --steps 20 --eval-steps 5 --no-cpu-baseline --no-secondary --no-quality > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_e.json').read().strip().splitlines()[-1]); e=d['eval']; print(d['eval_users_per_s'], e['ms_per_eval'], e['ms_score_topk_per_eval'], e['score_topk_equivalent_frac_of_f32_mfma_peak'], e['ms_per_eval_unseeded'], e['recall_at_20_after_timed_training'])"
