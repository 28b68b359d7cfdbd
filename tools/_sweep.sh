python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/t.log | tail -12
