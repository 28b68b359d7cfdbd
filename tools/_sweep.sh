python -m pytest tests/test_gpu_tile_engine.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert|FAILED" gpurun_out/t.log | tail -20
