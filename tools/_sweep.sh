python -m pytest tests/test_gpu_parity.py tests/test_gpu_carnn.py -m gpu -q --timeout 900 > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert|FAILED" gpurun_out/t.log | tail -20
