python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert|FAILED" gpurun_out/t.log | tail -10
(time python bench.py --shape x1) > gpurun_out/bench_x1.json 2> gpurun_out/bench_x1.err; tail -8 gpurun_out/bench_x1.err
python - <<EOP
import json
d=json.loads(open("gpurun_out/bench_x1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["eval_users_per_s"], d["eval"], d["config"])
print({k:(round(v["ms_per_step"],3), round(v.get("frac",0),3)) for k,v in d["kernels"].items()})
EOP
