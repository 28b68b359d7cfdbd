#!/bin/bash
# Per-phase cycle counts of te_head (a -DTE_HEAD_PROF build of the library: clock64 stamps between the kernel's barriers, summed per
# wave over all workgroups, printed at every training launch).  Run through gpurun AFTER building with
#   POI_HIPCC_FLAGS=-DTE_HEAD_PROF python -m poi_amd.build      (and rebuild without the flag afterwards)
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-quality --eval-steps 1 > gpurun_out/hp.json 2> gpurun_out/hp.err
grep "te_head prof" gpurun_out/hp.err | tail -4
