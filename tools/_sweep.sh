python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/t.log | tail -12
python bench.py --steps 30 --no-quality --no-secondary --no-cpu-baseline > gpurun_out/bench_b.json 2>/dev/null; python - <<EOP
import json
d=json.loads(open("gpurun_out/bench_b.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["eval_users_per_s"], d["eval"])
EOP
