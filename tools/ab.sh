#!/bin/bash
# usage: tools/ab.sh <filter> <dbg values...>   - bench one train step per POI_TE_DBG value, print matching kernels
flt=$1; shift
for d in "$@"; do
  POI_TE_DBG=$d python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eval 2>/dev/null | D=$d F=$flt python -c "
import sys,json,os
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j.get('kernels',{})
print('dbg', os.environ['D'], round(j['ms_per_step'],3), {n:round(v['ms_per_step'],3) for n,v in k.items() if os.environ['F'] in n})"
done
