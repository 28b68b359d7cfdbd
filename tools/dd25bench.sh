#!/bin/bash
# Short timing of the 1520-bin configuration on the GPU box (through gpurun): sequences/s, ms per epoch, per-kernel ms.
timeout 250 python bench.py --dd 25 --ud-km 38 --steps 20 --no-cpu-baseline --no-secondary --no-quality > gpurun_out/dd25.json 2> gpurun_out/dd25.err; python -c "
import json; d=json.loads(open('gpurun_out/dd25.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); k=d['kernels']
print({n: round(v['ms_per_step'],3) for n,v in k.items() if isinstance(v,dict) and 'ms_per_step' in v})"
