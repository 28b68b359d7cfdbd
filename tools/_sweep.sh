python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -x > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/t.log | tail -12
