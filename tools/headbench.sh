#!/bin/bash
# Short timing of the training launch on the GPU box (through gpurun): throughput, epoch time and the per-kernel regions.
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-secondary --no-quality --eval-steps 1 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_a.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']);
for k,v in d['kernels'].items(): print(k, round(v['ms_per_step'],3), round(1e3*v['avg_ms'],1), v.get('frac'))"
