#!/usr/bin/env python
"""Print the per-kernel table of a bench.py JSON line read from stdin."""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("B=%s  %d seq/s  %.2f ms/step  eval %s users/s  train %.1f TF" % (d["config"]["batch_users_per_launch"], d["value"], d["ms_per_step"],
      d["eval_users_per_s"] and round(d["eval_users_per_s"]), d["train_step_tflops"] or 0))
for k, v in d["kernels"].items():
    print("   %-12s %7.3f ms  %s" % (k, v["ms_per_step"], ("%.1f %s (%.0f%%)" % (v["achieved"], v["unit"], 100 * v["frac"])) if "achieved" in v else ""))
if d.get("eval"):
    print("   eval:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d["eval"].items()})
