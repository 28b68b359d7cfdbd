#!/bin/bash
# Rebuild libpoi_hip.so from the repo root (POI_HIPCC_FLAGS passes extra hipcc flags, e.g. -DTE_HEAD_PROF) and list the
# register / LDS / spill figures of the kernels whose names contain one of the arguments.
cd "$(dirname "$0")/.." || exit 1
python -c "
import sys; sys.path.insert(0, '.')
import importlib; b = importlib.import_module('point-of-interest-recommendation_amd.build'); b.build_lib(force=True, verbose=False)" 2>&1 | grep -iE "error" | head
