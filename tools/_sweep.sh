python -m pytest tests/test_gpu_multi.py -m gpu -q --timeout 900 -x -k emulated > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert|recall" gpurun_out/t.log | tail -12
