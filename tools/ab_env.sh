#!/bin/bash
# usage: tools/ab_env.sh VAR v1 v2 ...   - alternate runs of the training bench under VAR=v (twice each), print ms per epoch and the per-kernel split
var=$1; shift
for r in 1 2; do for v in "$@"; do echo "$var=$v"; env $var=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-eval --no-secondary --no-quality 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j.get('kernels',{})
print(round(j['value']), round(j['ms_per_step'],3), {n:round(v['ms_per_step'],3) for n,v in k.items() if isinstance(v,dict) and 'ms_per_step' in v})"; done; done
